"""Pins the CPU oracle directly against the only image the reference itself produced.

`fixtures/mdimg.png` is the reference's own render of scene1.ron with the defaults of src/bin/offline.rs
(1280x720, 1000 frames x 1 spp, depth 50).  Its seeds are unknown (rand::random(), mod.rs:128,242), so the
comparison is statistical: RMSE / PSNR of the 8-bit gamma-space images.  This script renders the same set-up
with the oracle (oracle/pt_oracle.cpp), the seed schedule the library uses when the caller passes no seeds
(frame k: mix of 0x00C0FFEE and k, restated below), for BOTH readings of the glass branch at
compute.wgsl:344 (the random number is always drawn / is skipped under total internal reflection), and writes

  tests/golden/mdimg_oracle_1000.png   the oracle's rgb8 image, always-draws reading (the default)
  tests/golden/mdimg_check.json        RMSE / PSNR of both readings against mdimg.png, how far the two
                                       readings are from each other, and an FNV-1a hash of the rgb8 bytes

The GPU test `test_oracle_image_of_the_reference_setup_byte_for_byte` renders the same 1000 frames through the
C ABI and demands the golden's bytes.  About five minutes on 8 cores:  python tests/golden/make_mdimg_check.py
"""
import json
import os
import sys
import time

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
W, H, FRAMES, BASE = 1280, 720, 1000, 0x00C0FFEE


def frame_seed(base, k):
    m = 0xFFFFFFFF
    x = (base ^ ((k * 0x9E3779B9 + 0x7F4A7C15) & m)) & m
    x ^= x >> 16
    x = (x * 0x85EBCA6B) & m
    x ^= x >> 13
    x = (x * 0xC2B2AE35) & m
    x ^= x >> 16
    return x


def fnv1a(data):
    h = 0xCBF29CE484222325
    for b in np.frombuffer(data, np.uint8).tolist():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def rmse(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


def main():
    threads = int(os.environ.get("ORC_THREADS", os.cpu_count() or 8))
    ref = np.asarray(Image.open(os.path.join(ROOT, "fixtures", "mdimg.png")).convert("RGB"))
    assert ref.shape == (H, W, 3)
    sc = orc.Scene.from_ron(os.path.join(ROOT, "fixtures", "scene1.ron"))
    seeds = [frame_seed(BASE, k) for k in range(FRAMES)]
    images, report = {}, {"width": W, "height": H, "frames": FRAMES, "base_seed": BASE, "readings": {}}
    for name, always in (("always_draws", 1), ("short_circuit", 0)):
        orc.lib().orc_set_glass_always_draws(always)
        t0 = time.time()
        acc, st = orc.render(sc, sc.camera(W, H), W, H, seeds, threads=threads)
        img = orc.postprocess(acc, FRAMES)
        images[name] = img
        e = rmse(img, ref)
        report["readings"][name] = {"rmse_8bit": e, "psnr_db": float(20 * np.log10(255.0 / e)), "rays": st["rays"],
                                    "fnv1a_rgb8": "%016x" % fnv1a(img.tobytes()), "seconds": round(time.time() - t0, 1)}
        print(name, report["readings"][name], flush=True)
    orc.lib().orc_set_glass_always_draws(1)
    a, b = images["always_draws"], images["short_circuit"]
    report["between_readings"] = {"rmse_8bit": rmse(a, b),
                                  "pixels_differing": int(np.count_nonzero((a != b).any(axis=2))),
                                  "max_abs_diff_8bit": int(np.abs(a.astype(int) - b.astype(int)).max())}
    Image.fromarray(a, "RGB").save(os.path.join(OUT, "mdimg_oracle_1000.png"), optimize=True)
    with open(os.path.join(OUT, "mdimg_check.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report["between_readings"]))


if __name__ == "__main__":
    main()
