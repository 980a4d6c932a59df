"""Node visits and sphere tests per ray of the default tree on a scene, counted by the CPU harness of the
kernels' per-lane code (development aid; no GPU).  usage: python tools/tree_quality.py [scene.ron] [W H frames]"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pa
from oracle import orc

here = os.path.join(ROOT, "tests", "emul")
so = os.path.join(here, "libpt_emul.so")
srcs = [os.path.join(here, "pt_emul.cpp")] + [os.path.join(ROOT, "pathtracer_amd", "csrc", f) for f in ("scene_pack.cpp", "bvh.cpp")]
subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", "-o", so] + srcs)
L = C.CDLL(so)
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "fixtures", "scene1.ron")
W, H, F = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (192, 108, 4)
sc = pa.load_ron_scene(path)
cam = np.ascontiguousarray(orc.Scene.from_ron(path).camera(W, H), np.float32)
seeds = np.arange(1, F + 1, dtype=np.uint32) * 7919
acc = np.zeros((H, W, 4), np.float32)
cnt = (C.c_uint64 * 10)()
info = (C.c_uint32 * 4)()
vp, sz = C.c_void_p, C.c_size_t
r = L.emul_render(vp(sc.spheres.ctypes.data), sz(len(sc.spheres)), vp(sc.lights.ctypes.data), sz(len(sc.lights)),
                  vp(sc.lambertians.ctypes.data), sz(len(sc.lambertians)), vp(sc.metals.ctypes.data), sz(len(sc.metals)),
                  vp(sc.glass.ctypes.data), sz(len(sc.glass)), C.c_int(0), vp(None), sz(0), vp(cam.ctypes.data),
                  C.c_uint32(W), C.c_uint32(H), C.c_int32(1), C.c_int32(50), C.c_int(1), C.c_uint32(F),
                  vp(seeds.ctypes.data), vp(acc.ctypes.data), C.c_uint32(0), C.c_uint32(W), C.c_uint32(0), C.c_uint32(H), cnt, info)
assert r == 0
nodes, tests, rays = cnt[0], cnt[1], cnt[8]
print(f"nodes {info[0]} depth {info[1]} format {info[3]}: {nodes / rays:.3f} node visits + {tests / rays:.3f} sphere tests per ray "
      f"(1.0 n + 0.8 t = {(nodes + 0.8 * tests) / rays:.3f}); image checksum {int(acc.view(np.uint32).sum(dtype=np.uint64))}")
