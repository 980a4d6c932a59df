"""Development probe: one random scene (tests/conftest.py: random_scene) on the GPU against the oracle."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import pathtracer_amd as pa
from oracle import orc
from conftest import random_scene, oracle_scene
orc.build()
seed = int(sys.argv[1])
W, H = 48, 32
sc = random_scene(pa, seed, duplicates=False)
osc = oracle_scene(orc, sc)
print("spheres", len(sc.spheres))
for depth in [0, 1, 2, 3, 4, 5, 6, 8, 10, 20, 50]:
    ref, st = orc.render(osc, osc.camera(W, H), W, H, [seed, 7], samples=1, depth=depth, threads=16)
    out = []
    for kern in ("sched", "lockstep"):
        os.environ["MI355PT_KERNEL"] = kern
        with pa.Renderer(sc, pa.Settings(1, depth), W, H, bvh_mode=pa.BVH_LBVH) as r:
            r.render(seeds=[seed, 7])
            acc = r.read_accum()
            d = (acc.view(np.uint32) != ref.view(np.uint32)).any(axis=2)
            out.append((kern, int(d.sum()), r.stats()["rays"] - st["rays"], np.argwhere(d)[:3].tolist(), r.stats()["bvh_nodes"], r.stats()["bvh_depth"]))
    print("depth", depth, out)
