"""Per-frame values of one texel of a soak scene for every tree, against the oracle (development aid).
usage: python tools/soak_pixel.py <seed> <px> <py>     (SOAK_W, SOAK_H, SOAK_FRAMES as tools/soak.py)"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import pathtracer_amd as pa
from oracle import orc
from conftest import random_scene, oracle_scene
orc.build()
seed, px, py = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
W, H = int(os.environ.get("SOAK_W", 48)), int(os.environ.get("SOAK_H", 32))
samples, depth = (1, 50) if seed % 4 else (2, 7)
nf = int(os.environ.get("SOAK_FRAMES", 2))
seeds = [seed, 7] + [seed * 31 + k for k in range(nf - 2)]
sc = random_scene(pa, seed, duplicates=False)
osc = oracle_scene(orc, sc)
cam = osc.camera(W, H)
for f, s in enumerate(seeds):
    ref, st = orc.render(osc, cam, W, H, [s], samples=samples, depth=depth, crop=(px, px + 1, py, py + 1))
    row = [f, s, ref[py, px, :3].tolist()]
    for mode in (pa.BVH_SAH, pa.BVH_LBVH):
        with pa.Renderer(sc, pa.Settings(samples, depth), W, H, bvh_mode=mode) as r:
            r.render(seeds=[s])
            a = r.read_accum()[py, px, :3]
            row.append("same" if (a.view(np.uint32) == ref[py, px, :3].view(np.uint32)).all() else a.tolist())
    print(*row, flush=True)
