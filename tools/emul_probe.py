"""Development probe (CPU harness): for random scenes, the tree walk against brute force over all spheres and
against brute force culled by each sphere's own padded box; the latter logs every root such a box would hide.
usage: python tools/emul_probe.py <seed> [<seed> ...]   (needs tests/emul/libpt_emul.so: run the CPU tests once)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import numpy as np
import pathtracer_amd as pa
from oracle import orc
from conftest import random_scene, oracle_scene
orc.build()
L = C.CDLL(os.path.join(ROOT, "tests", "emul", "libpt_emul.so"))
L.emul_error.restype = C.c_char_p
def render(sc, cam, W, H, seeds, samples, depth, topo=0):
    acc = np.zeros((H, W, 4), np.float32); acc[..., 3] = 1
    seeds = np.ascontiguousarray(seeds, np.uint32); cam = np.ascontiguousarray(cam, np.float32)
    cnt = (C.c_uint64 * 10)(); info = (C.c_uint32 * 4)()
    vp, sz = C.c_void_p, C.c_size_t
    r = L.emul_render(vp(sc.spheres.ctypes.data), sz(len(sc.spheres)), vp(sc.lights.ctypes.data), sz(len(sc.lights)),
                      vp(sc.lambertians.ctypes.data), sz(len(sc.lambertians)), vp(sc.metals.ctypes.data), sz(len(sc.metals)),
                      vp(sc.glass.ctypes.data), sz(len(sc.glass)), C.c_int(topo), vp(None), sz(0), vp(cam.ctypes.data),
                      C.c_uint32(W), C.c_uint32(H), C.c_int32(samples), C.c_int32(depth), C.c_int(1), C.c_uint32(len(seeds)),
                      vp(seeds.ctypes.data), vp(acc.ctypes.data), C.c_uint32(0), C.c_uint32(W), C.c_uint32(0), C.c_uint32(H), cnt, info)
    assert r == 0, L.emul_error()
    return acc, int(cnt[8])
DUP = os.environ.get('PROBE_DUP', '0') == '1'
for seed in [int(a) for a in sys.argv[1:]]:
    W, H = 48, 32
    samples, depth = (1, 50) if seed % 4 else (2, 7)
    sc = random_scene(pa, seed, duplicates=DUP)
    osc = oracle_scene(orc, sc)
    cam = osc.camera(W, H)
    ref, st = orc.render(osc, cam, W, H, [seed, 7], samples=samples, depth=depth, threads=8)
    L.emul_set_brute_force(0)
    tree, rays_t = render(sc, cam, W, H, [seed, 7], samples, depth)
    L.emul_set_brute_force(1)
    brute, rays_b = render(sc, cam, W, H, [seed, 7], samples, depth)
    L.emul_set_brute_force(2)
    boxed, rays_x = render(sc, cam, W, H, [seed, 7], samples, depth)
    d3 = (boxed.view(np.uint32) != ref.view(np.uint32)).any(axis=2)
    print("own-box cull vs oracle:", int(d3.sum()), np.argwhere(d3)[:3].tolist())
    d1 = (tree.view(np.uint32) != ref.view(np.uint32)).any(axis=2)
    d2 = (brute.view(np.uint32) != ref.view(np.uint32)).any(axis=2)
    print(seed, "tree vs oracle:", int(d1.sum()), "brute force vs oracle:", int(d2.sum()), np.argwhere(d2)[:3].tolist(), "rays", st["rays"], rays_t, rays_b)
