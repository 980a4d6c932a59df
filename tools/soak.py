"""Development soak: random scenes (tests/conftest.py) x three trees on the GPU against the oracle.
usage: python tools/soak.py <count> [<first seed>]
SOAK_FRAMES=n (default 2): frames per render; with n >= 10 the first 9 go in one call (which measures the tile order)
and the rest in a second one, so that the sky-tile split of the accumulation kernel is exercised too;
SOAK_QUEUE=1: the rest is queued two frames per call without waiting, so that launches add their predecessor's
results (with MI355PT_XCHG=1 MI355PT_BLOCK=1024, which force the kernel instance that does that at this size);
SOAK_W, SOAK_H: image size (default 48 x 32)."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import pathtracer_amd as pa
from oracle import orc
from conftest import random_scene, oracle_scene, bits_equal
orc.build()
bad = unexplained = 0
start = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
for seed in range(start, start + int(sys.argv[1])):
    W, H = int(os.environ.get("SOAK_W", 48)), int(os.environ.get("SOAK_H", 32))
    samples, depth = (1, 50) if seed % 4 else (2, 7)
    for dup, modes in ((True, (pa.BVH_SAH, pa.BVH_REFERENCE, pa.BVH_LBVH)), (False, (pa.BVH_LBVH,))):
        sc = random_scene(pa, seed, duplicates=dup)
        osc = oracle_scene(orc, sc)
        nf = int(os.environ.get("SOAK_FRAMES", 2))
        seeds = [seed, 7] + [seed * 31 + k for k in range(nf - 2)]
        ref, st = orc.render(osc, osc.camera(W, H), W, H, seeds, samples=samples, depth=depth, threads=16)
        for mode in modes:
            with pa.Renderer(sc, pa.Settings(samples, depth), W, H, bvh_mode=mode) as r:
                if nf >= 10:
                    r.render(seeds=seeds[:9])
                    if os.environ.get("SOAK_QUEUE") == "1":
                        for k in range(9, nf, 2):
                            r.render(seeds=seeds[k:k + 2], wait=False)
                    else:
                        r.render(seeds=seeds[9:])
                else:
                    r.render(seeds=seeds)
                if not bits_equal(ref, r.read_accum()) or r.stats()["rays"] != st["rays"]:
                    bad += 1
                    # paths that cross a thousand units of glass come back at the small spheres with roots no padded
                    # box contains: what is hit then depends on the boxes a walk tests, in the reference as here
                    known = bool(((sc.spheres["rad"] >= 100) & (sc.spheres["mat_type"] == 3)).any()) and mode != pa.BVH_REFERENCE
                    unexplained += 0 if known else 1
                    print("MISMATCH seed", seed, "mode", mode, "(glass sphere of radius >= 100: tree-dependent class)" if known else "UNEXPLAINED", flush=True)
print("done, mismatches:", bad, "unexplained:", unexplained)
