"""Development soak: random scenes (tests/conftest.py) x three trees on the GPU against the oracle.
usage: python tools/soak.py <count> [<first seed>]"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import pathtracer_amd as pa
from oracle import orc
from conftest import random_scene, oracle_scene, bits_equal
orc.build()
bad = 0
start = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
for seed in range(start, start + int(sys.argv[1])):
    W, H = 48, 32
    samples, depth = (1, 50) if seed % 4 else (2, 7)
    for dup, modes in ((True, (pa.BVH_SAH, pa.BVH_REFERENCE)), (False, (pa.BVH_LBVH,))):
        sc = random_scene(pa, seed, duplicates=dup)
        osc = oracle_scene(orc, sc)
        ref, st = orc.render(osc, osc.camera(W, H), W, H, [seed, 7], samples=samples, depth=depth, threads=16)
        for mode in modes:
            with pa.Renderer(sc, pa.Settings(samples, depth), W, H, bvh_mode=mode) as r:
                r.render(seeds=[seed, 7])
                if not bits_equal(ref, r.read_accum()) or r.stats()["rays"] != st["rays"]:
                    bad += 1
                    print("MISMATCH seed", seed, "mode", mode, flush=True)
print("done, mismatches:", bad)
