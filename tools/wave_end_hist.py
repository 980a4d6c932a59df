"""When the waves of a path-kernel launch end (development aid).  Needs a library built with
-DPT_WAVE_END_HIST=<shift> (bucket = 2^shift ticks of 10 ns) and MI355PT_LIB pointing at it.
usage: MI355PT_LIB=pathtracer_amd/libmi355pt_hist.so python tools/wave_end_hist.py FRAMES SHIFT"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pathtracer_amd as pa
F, shift = int(sys.argv[1]), int(sys.argv[2])
sc = pa.load_ron_scene("fixtures/scene1.ron")
L = pa.lib()
L.mi355pt_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t]
with pa.Renderer(sc, pa.Settings(1, 50), 1920, 1080) as r:
    r.render(F); r.render(F)           # tile order measured
    for rep in range(2):
        r.clear()                       # zeroes the counters
        r.render(F)
        buf = (C.c_uint64 * 128)()
        L.mi355pt_debug_counters(r._ctx, buf, 128)
        h = np.array(buf[32:96], np.int64)
        ms = (1 << shift) * 1e-5
        # rotate so that the histogram ends with its last non-empty bucket before the longest empty run
        empty = h == 0
        best, run, start = 0, 0, 0
        for i in range(128):
            if empty[i % 64]:
                run += 1
                if run > best: best, start = run, (i - run + 1) % 64
            else: run = 0
        order = [(start + best + k) % 64 for k in range(64 - best)]
        print(f"{F} frames, kernel {r.stats()['render_ms']:.2f} ms, bucket {ms:.3f} ms; waves ending per bucket, last bucket last:")
        print(" ".join(str(int(h[i])) for i in order))
