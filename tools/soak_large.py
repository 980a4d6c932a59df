"""Development soak for trees that stay in global memory: random clouds of 2 000 - 30 000 spheres on the GPU
(default tree and device-built tree, whatever kernels MI355PT_BLOCK selects) against the oracle, bit for bit.
usage: python tools/soak_large.py <count> [<first seed>]"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import pathtracer_amd as pa
from pathtracer_amd.renderer import SPHERE_DT, VEC16_DT, CameraSettings, Scene
from oracle import orc
from conftest import oracle_scene, bits_equal
orc.build()

def cloud(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(2000, 30000))
    sp = np.zeros(n, SPHERE_DT)
    ext = float(rng.choice([10.0, 30.0, 80.0]))
    sp["pos"] = rng.uniform((-ext, 0.0, -ext), (ext, ext / 4, ext), (n, 3)).astype(np.float32)
    sp["rad"] = rng.choice([0.05, 0.2, 0.4, 1.0], n, p=[.3, .4, .25, .05]).astype(np.float32)
    sp[0] = ((0.0, -1000.0, 0.0), 1000.0, 1, 0, (0, 0))
    counts = rng.integers(1, 6, 4)
    sp["mat_type"] = rng.integers(0, 4, n)
    sp["mat_type"][rng.random(n) < 0.6] = 1
    sp["mat_index"] = [rng.integers(0, counts[t]) for t in sp["mat_type"]]
    lights = np.zeros(counts[0], VEC16_DT); lights["v"] = rng.uniform(0.5, 6.0, (counts[0], 3)).astype(np.float32)
    lam = np.zeros(counts[1], VEC16_DT); lam["v"] = rng.uniform(0.0, 1.0, (counts[1], 3)).astype(np.float32)
    met = np.zeros(counts[2], VEC16_DT); met["v"] = rng.uniform(0.2, 1.0, (counts[2], 3)).astype(np.float32)
    met["w"] = rng.choice([0.0, 0.1, 0.5, 1.0], counts[2]).astype(np.float32)
    gla = rng.choice([0.7, 1.0, 1.01, 1.5, 2.4], counts[3]).astype(np.float32)
    cam = CameraSettings(tuple(rng.uniform((-ext, 1, -ext), (ext, ext / 3, ext)).astype(np.float32)),
                         np.float32(rng.uniform(-180, 180)), np.float32(rng.uniform(-40, 10)), np.float32(rng.choice([20.0, 45.0, 70.0])))
    return Scene(cam, sp, lights, lam, met, gla)

bad = 0
start = int(sys.argv[2]) if len(sys.argv) > 2 else 70000
for seed in range(start, start + int(sys.argv[1])):
    sc = cloud(seed)
    osc = oracle_scene(orc, sc)
    W, H = 64, 48
    ref, st = orc.render(osc, osc.camera(W, H), W, H, [seed, 7], threads=16)
    for mode in (pa.BVH_SAH, pa.BVH_LBVH):
        with pa.Renderer(sc, pa.Settings(1, 50), W, H, bvh_mode=mode) as r:
            r.render(seeds=[seed, 7])
            s = r.stats()
            d = int((r.read_accum().view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum())
            if d or s["rays"] != st["rays"] or s["scene_in_lds"]:
                bad += 1
                print("MISMATCH seed", seed, "mode", mode, "pixels", d, "rays", s["rays"], st["rays"], "in_lds", s["scene_in_lds"], flush=True)
    print("seed", seed, "spheres", len(sc.spheres), flush=True)
print("done, mismatches:", bad)
