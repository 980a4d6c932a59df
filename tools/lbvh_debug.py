"""A soak scene's GPU-built tree read back and walked on the CPU harness, frame by frame for one texel (development
aid; run via gpurun).  usage: python tools/lbvh_debug.py <seed> <px> <py> <dup 0|1>   (SOAK_W, SOAK_H, SOAK_FRAMES as
tools/soak.py; PT_EMUL_TRACE=1 lists every ray of the harness's paths on stderr)"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import pathtracer_amd as pa
from oracle import orc
import conftest
from conftest import random_scene, oracle_scene, device_scene_blob
orc.build()
emul = conftest.emul.__wrapped__()
seed, px, py, dup = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
W, H = int(os.environ.get("SOAK_W", 48)), int(os.environ.get("SOAK_H", 32))
samples, depth = (1, 50) if seed % 4 else (2, 7)
nf = int(os.environ.get("SOAK_FRAMES", 2))
seeds = [seed, 7] + [seed * 31 + k for k in range(nf - 2)]
sc = random_scene(pa, seed, duplicates=bool(dup))
osc = oracle_scene(orc, sc)
cam = osc.camera(W, H)
crop = (px, px + 1, py, py + 1)
with pa.Renderer(sc, pa.Settings(samples, depth), W, H, bvh_mode=pa.BVH_LBVH) as r:
    blob, layout = device_scene_blob(r)
emul.set_blob(blob, layout)
print("tree check:", emul.emul_check_blob_tree())
for f, s in enumerate(seeds):
    ref, _ = orc.render(osc, cam, W, H, [s], samples=samples, depth=depth, crop=crop)
    emul.set_blob(blob, layout)
    print("-- frame", f, "device tree", file=sys.stderr, flush=True)
    a, c, _ = emul.render(sc, cam, W, H, [s], samples=samples, depth=depth, crop=crop)
    same = (a[py, px].view(np.uint32) == ref[py, px].view(np.uint32)).all()
    print("frame", f, "seed", s, "oracle", ref[py, px, :3], "device tree", "same" if same else a[py, px, :3], flush=True)
    if not same:
        emul.set_blob(None)
        print("-- frame", f, "host tree", file=sys.stderr, flush=True)
        emul.render(sc, cam, W, H, [s], samples=samples, depth=depth, crop=crop)
print(sc.spheres)
