"""One soak scene under the current environment, with the differing texels listed (development aid).
usage: python tools/soak_one.py <seed> [queue]"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import pathtracer_amd as pa
from oracle import orc
from conftest import random_scene, oracle_scene
orc.build()
seed = int(sys.argv[1])
queue = len(sys.argv) > 2
W, H = int(os.environ.get("SOAK_W", 48)), int(os.environ.get("SOAK_H", 32))
samples, depth = (1, 50) if seed % 4 else (2, 7)
nf = int(os.environ.get("SOAK_FRAMES", 2))
seeds = [seed, 7] + [seed * 31 + k for k in range(nf - 2)]
for dup, modes in ((True, (pa.BVH_SAH, pa.BVH_REFERENCE, pa.BVH_LBVH)), (False, (pa.BVH_SAH, pa.BVH_REFERENCE, pa.BVH_LBVH))):
    sc = random_scene(pa, seed, duplicates=dup)
    osc = oracle_scene(orc, sc)
    ref, st = orc.render(osc, osc.camera(W, H), W, H, seeds, samples=samples, depth=depth, threads=16)
    for mode in modes:
        with pa.Renderer(sc, pa.Settings(samples, depth), W, H, bvh_mode=mode) as r:
            if nf >= 10:
                r.render(seeds=seeds[:9])
                if queue:
                    for k in range(9, nf, 2):
                        r.render(seeds=seeds[k:k + 2], wait=False)
                else:
                    r.render(seeds=seeds[9:])
            else:
                r.render(seeds=seeds)
            a = r.read_accum()
            s = r.stats()
            d = (a.view(np.uint32) != ref.view(np.uint32)).any(axis=2)
            ys, xs = np.nonzero(d)
            print("dup", dup, "seed", seed, "spheres", len(sc.spheres), "mode", mode, "samples", samples, "differ", int(d.sum()), "rays", s["rays"], st["rays"],
                  "list_tiles", s["list_tiles"], "sky_tiles", s["sky_tiles"], "launches", s["launches"],
                  [(int(x), int(y), a[y, x, :3].tolist(), ref[y, x, :3].tolist()) for y, x in list(zip(ys, xs))[:4]], flush=True)
