"""Regenerates tests/golden/*.npz|json from the CPU oracle (run from the repo root).

The reference cannot be executed in this pipeline and ships no vectors for the compute path, so
these pins come from the oracle restatement: they guard it (and the GPU path checked against it)
against drift across compilers, flags and edits.  `python tests/golden/make_golden.py`
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
SEEDS = [0x00C0FFEE, 0x12345678, 0x9E3779B9, 0xDEADBEEF]


def fnv1a(data):
    h = 0xCBF29CE484222325
    for b in data:
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def main():
    import ctypes as C
    L = orc.lib()
    meta = {"seeds": SEEDS, "width": 64, "height": 36, "depth": 50, "samples": 1}
    arrays = {}
    # RNG: 16 draws for 8 seeds
    rng_seeds = [1, 2, 0x00C0FFEE, 0xFFFFFFFF, 0x80000000, 12345, 0x9E3779B9, 0x7FFFFFFF]
    draws = np.zeros((8, 16), np.uint32)
    for i, s in enumerate(rng_seeds):
        st = C.c_uint32(s)
        for j in range(16):
            draws[i, j] = L.orc_randu(C.byref(st))
    arrays["rng_seeds"] = np.array(rng_seeds, np.uint32)
    arrays["rng_draws"] = draws
    # pixel hash on a 64x36 and a 1920x1080 grid (corners + a diagonal)
    pts = [(x, y, 64, 36) for x in (0, 1, 31, 63) for y in (0, 17, 35)] + \
          [(x, y, 1920, 1080) for x in (0, 959, 960, 1919) for y in (0, 539, 540, 1079)]
    arrays["hash_points"] = np.array(pts, np.uint32)
    arrays["hash_values"] = np.array([L.orc_pixel_seed(0, *p) for p in pts], np.uint32)
    arrays["sin_x"] = np.linspace(-91.3, 91.3, 257).astype(np.float32)
    arrays["sin_y"] = np.array([L.orc_det_sin(float(x)) for x in arrays["sin_x"]], np.float32)
    for name in ("scene1", "scene2", "scene3"):
        sc = orc.Scene.from_ron(os.path.join(ROOT, "fixtures", name + ".ron"))
        cam = sc.camera(64, 36)
        arrays[name + "_camera_16x9"] = cam
        acc, st = orc.render(sc, cam, 64, 36, SEEDS)
        arrays[name + "_accum"] = acc
        meta[name + "_stats"] = st
        if name == "scene1":
            meta["scene1_bvh_fnv1a"] = "%016x" % fnv1a(sc.nodes.tobytes())
            meta["scene1_bvh_nodes"] = int(len(sc.nodes))
            cam720 = sc.camera(1280, 720)
            _, st0 = orc.render(sc, cam720, 1280, 720, [SEEDS[0]], depth=0, threads=8)
            meta["scene1_primary_1280x720"] = st0
        else:
            arrays[name + "_bvh"] = np.frombuffer(sc.nodes.tobytes(), np.uint8)
        # multi-sample frames and a shallow depth, smaller image
        acc2, st2 = orc.render(sc, sc.camera(40, 30), 40, 30, SEEDS[:2], samples=3, depth=4)
        arrays[name + "_accum_s3_d4"] = acc2
        arrays[name + "_camera_4x3"] = sc.camera(40, 30)
    np.savez_compressed(os.path.join(OUT, "golden.npz"), **arrays)
    with open(os.path.join(OUT, "golden.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
