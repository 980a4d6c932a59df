"""Soak at realistic sizes without the CPU oracle: random scenes (tests/conftest.py) rendered with the default tree,
the GPU-built tree and in the reference-walk mode (which runs exactly the reference's sphere tests and is checked
against the oracle elsewhere); long launches, so the exchange kernel, candidate lists, sky tiles and queued launches
that add their predecessor's results all run as in production.  Scenes whose huge ground sphere came out as glass
get a lambertian one: paths that cross a thousand units of glass are the known tree-dependent class.
usage: python tools/soak_gpu.py <count> [<first seed>]   (SOAK_W, SOAK_H, SOAK_FRAMES: default 1280 x 720 x 96)"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import pathtracer_amd as pa
from conftest import random_scene, bits_equal
W, H = int(os.environ.get("SOAK_W", 1280)), int(os.environ.get("SOAK_H", 720))
F = int(os.environ.get("SOAK_FRAMES", 96))
start = int(sys.argv[2]) if len(sys.argv) > 2 else 90000
bad = 0
for seed in range(start, start + int(sys.argv[1])):
    sc = random_scene(pa, seed, duplicates=True)
    big_glass = (sc.spheres["rad"] >= 100) & (sc.spheres["mat_type"] == 3)
    sc.spheres["mat_type"][big_glass] = 1
    sc.spheres["mat_index"][big_glass] = 0
    seeds = np.array([pa.frame_seed(seed, k) for k in range(F)], np.uint32)
    out = {}
    for mode in (pa.BVH_REFERENCE, pa.BVH_SAH, pa.BVH_LBVH):
        with pa.Renderer(sc, pa.Settings(1, 50), W, H, bvh_mode=mode) as r:
            r.render(seeds=seeds[:16])                       # measures the tile order
            rest = (F - 16) // 2
            r.render(seeds=seeds[16:16 + rest], wait=False)  # queued: the second adds the first's results
            r.render(seeds=seeds[16 + rest:], wait=False)
            out[mode] = (r.read_accum(), r.stats())
    ref, rs = out[pa.BVH_REFERENCE]
    for mode in (pa.BVH_SAH, pa.BVH_LBVH):
        a, s = out[mode]
        if not bits_equal(ref, a) or s["rays"] != rs["rays"]:
            bad += 1
            d = (a.view(np.uint32) != ref.view(np.uint32)).any(axis=2)
            print("MISMATCH seed", seed, "mode", mode, "texels", int(d.sum()), "rays", s["rays"], rs["rays"], np.argwhere(d)[:3].tolist(), flush=True)
    if (seed - start) % 20 == 19:
        s = out[pa.BVH_SAH][1]
        print("..", seed - start + 1, "scenes; last:", len(sc.spheres), "spheres, exchange", s["exchange_on"], "sky tiles", s["sky_tiles"],
              "list tiles", s["list_tiles"], "launches", s["launches"], flush=True)
print("done, mismatches:", bad)
