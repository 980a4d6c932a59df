"""Per-call cost of the interactive pattern of main.rs:119-176: one frame per redraw, accumulator kept.
usage: python tools/interactive.py [W H] (run via gpurun)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracer_amd as pa
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
sc = pa.load_ron_scene("fixtures/scene1.ron")
with pa.Renderer(sc, pa.Settings(1, 50), W, H) as r:
    r.render(16)
    for mode in ("blocking", "queued"):
        r.clear()
        t0 = time.perf_counter()
        n = 200
        for _ in range(n):
            r.render(1, wait=(mode == "blocking"))
        r.wait()
        dt = time.perf_counter() - t0
        s = r.stats()
        print(f"{W}x{H} {mode}: {dt / n * 1e3:.3f} ms per frame call, kernel {s['render_ms'] / n:.3f} ms, "
              f"accumulation {s['resolve_ms'] / n:.3f} ms, {s['paths'] / dt / 1e6:.0f} Mpaths/s", flush=True)
