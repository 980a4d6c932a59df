"""Development probe: the 1 M-sphere plane at 3840x2160 x 256 frames on the GPU against oracle crops (one below
the horizon, one across it): how many accumulator texels differ, where, by how much.  Run via gpurun."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import conftest
import pathtracer_amd as pa
from oracle import orc
from pathtracer_amd import synth
N = int(os.environ.get("PROBE_SPHERES", 1_000_000))
F = int(os.environ.get("PROBE_FRAMES", 256))
sc = synth.plane_scene(N)
osc = conftest.oracle_scene(orc, sc)
W, H = 3840, 2160
seeds = np.array([pa.frame_seed(0x00C0FFEE, k) for k in range(F)], np.uint32)
with pa.Renderer(sc, pa.Settings(1, 50), W, H) as r:
    r.render(seeds=seeds)
    a = r.read_accum()
    print("gpu", r.stats()["render_ms"], "ms", flush=True)
for name, crop in (("below", (1900, 2028, 1800, 1864)), ("horizon", (1900, 1916, 600, 696))):
    x0, x1, y0, y1 = crop
    t = time.time()
    ref, st = orc.render(osc, osc.camera(W, H), W, H, seeds, crop=crop, threads=16)
    dt = time.time() - t
    A, R = a[y0:y1, x0:x1], ref[y0:y1, x0:x1]
    differ = (A.view(np.uint32) != R.view(np.uint32)).any(axis=2)
    rel = np.abs(A[..., :3] - R[..., :3]) / np.maximum(1e-3, np.abs(R[..., :3]))
    rows = np.unique(np.argwhere(differ)[:, 0]) + y0
    print(name, crop, f"oracle {dt:.1f}s", "differ", int(differ.sum()), "of", differ.size, "rows", rows.tolist()[:40],
          "max rel", float(rel.max()), "visits/ray", st["interior_visits"] / st["rays"], flush=True)
