"""C5 at its stated size on one GPU: the 10 M-sphere plane at 3840x2160, 16 384 frames.  Once in one context, once
as the eight-way frame partition of the configuration (eight contexts on this one device, mi355pt_reduce to the
root: on an eight-GPU node each context would sit on its own device and the sum would go through RCCL).  Prints the
two times and how far the two accumulators are apart (only the order of the float additions differs).
usage: python tools/c5_full.py [frames] [ranks]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pathtracer_amd as pa
from pathtracer_amd import synth

F = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
W, H = 3840, 2160
sc = synth.plane_scene(10_000_000)
seeds = np.array([pa.frame_seed(0x00C0FFEE, k) for k in range(F)], np.uint32)
t0 = time.perf_counter()
with pa.Renderer(sc, pa.Settings(1, 50), W, H, bvh_mode=pa.BVH_LBVH) as r:
    r.render(seeds=seeds)
    one, st = r.read_accum(), r.stats()
t_one = time.perf_counter() - t0
print(f"one context: {F} frames, {st['launches']} launches, kernels {st['render_ms'] + st['resolve_ms']:.0f} ms, "
      f"{st['rays'] / (st['render_ms'] + st['resolve_ms']) / 1e3:.0f} Mrays/s, wall {t_one:.1f} s", flush=True)
t0 = time.perf_counter()
rs = [pa.Renderer(sc, pa.Settings(1, 50), W, H, bvh_mode=pa.BVH_LBVH) for _ in range(N)]
try:
    for k, r in enumerate(rs):
        first, count = pa.partition_frames(F, N, k)
        r.render(seeds=seeds[first:first + count], wait=False)
    pa.reduce(rs, root=0)
    acc, s0 = rs[0].read_accum(), rs[0].stats()
    t_part = time.perf_counter() - t0
    rel = np.abs(acc[..., :3] - one[..., :3]) / np.maximum(1.0, np.abs(one[..., :3]))
    print(f"{N} contexts: samples {rs[0].samples()}, paths {s0['paths']} (one context {st['paths']}), rays equal {s0['rays'] == st['rays']}, "
          f"wall {t_part:.1f} s; relative difference max {rel.max():.2e}, rms {np.sqrt(np.mean(rel.astype(np.float64) ** 2)):.2e}; "
          f"alpha all 1: {bool((acc[..., 3] == 1.0).all())}", flush=True)
finally:
    for r in rs:
        r.close()
