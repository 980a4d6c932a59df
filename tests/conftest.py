import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FIXTURES = os.path.join(ROOT, "fixtures")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    # never initialises HIP in this process
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as o
    o.build()
    return o


@pytest.fixture(scope="session")
def pa():
    import pathtracer_amd
    pathtracer_amd.lib()
    return pathtracer_amd


@pytest.fixture(scope="session")
def emul():
    """CPU harness of the kernels' per-lane code (tests/emul/pt_emul.cpp); needs g++."""
    import ctypes as C
    here = os.path.join(ROOT, "tests", "emul")
    so = os.path.join(here, "libpt_emul.so")
    srcs = [os.path.join(here, "pt_emul.cpp"), os.path.join(ROOT, "pathtracer_amd", "csrc", "scene_pack.cpp"),
            os.path.join(ROOT, "pathtracer_amd", "csrc", "bvh.cpp")]
    deps = srcs + [os.path.join(ROOT, "pathtracer_amd", "csrc", f) for f in ("pt_device.h", "pt_types.h", "bvh.h",
                                                                           "scene_pack.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared",
                               "-o", so] + srcs)
    L = C.CDLL(so)
    L.emul_error.restype = C.c_char_p
    L.emul_pixel_hash.restype = C.c_uint32
    L.emul_pixel_hash.argtypes = [C.c_uint32] * 4
    L.emul_det_sin.restype = C.c_float
    L.emul_det_sin.argtypes = [C.c_float]

    def render(sc, cam, W, H, seeds, samples=1, depth=50, ref_topo=0, ref_nodes=None, crop=None, glass=1):
        acc = np.zeros((H, W, 4), np.float32)
        acc[..., 3] = 1
        seeds = np.ascontiguousarray(seeds, np.uint32)
        cam = np.ascontiguousarray(cam, np.float32)
        cnt = (C.c_uint64 * 10)()
        info = (C.c_uint32 * 4)()
        x0, x1, y0, y1 = crop if crop else (0, W, 0, H)
        rn = ref_nodes.ctypes.data if ref_nodes is not None else None
        nrn = len(ref_nodes) if ref_nodes is not None else 0
        vp, sz = C.c_void_p, C.c_size_t
        r = L.emul_render(vp(sc.spheres.ctypes.data), sz(len(sc.spheres)), vp(sc.lights.ctypes.data),
                          sz(len(sc.lights)), vp(sc.lambertians.ctypes.data), sz(len(sc.lambertians)),
                          vp(sc.metals.ctypes.data), sz(len(sc.metals)), vp(sc.glass.ctypes.data), sz(len(sc.glass)),
                          C.c_int(ref_topo), vp(rn), sz(nrn), vp(cam.ctypes.data), C.c_uint32(W), C.c_uint32(H),
                          C.c_int32(samples), C.c_int32(depth), C.c_int(glass), C.c_uint32(len(seeds)),
                          vp(seeds.ctypes.data), vp(acc.ctypes.data), C.c_uint32(x0), C.c_uint32(x1), C.c_uint32(y0),
                          C.c_uint32(y1), cnt, info)
        if r != 0:
            raise RuntimeError(L.emul_error().decode())
        names = ["nodes", "tests", "accepted", "b_light", "b_lambertian", "b_metal", "b_glass", "b_sky", "rays",
                 "paths"]
        return acc, dict(zip(names, [int(c) for c in cnt])), [int(i) for i in info]

    L.render = render
    return L


def oracle_scene(orc, scene, **kw):
    """pathtracer_amd.Scene (numpy blobs) -> oracle Scene with its own reference-topology tree."""
    cam = scene.camera
    blobs = {"camera": (np.array(cam.pos, np.float32), np.float32(cam.yaw), np.float32(cam.pitch),
                        np.float32(cam.vfov)),
             "spheres": scene.spheres, "lights": scene.lights, "lambertians": scene.lambertians,
             "metals": scene.metals, "glass": scene.glass}
    return orc.Scene(blobs, **kw)


def bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a, np.float32).view(np.uint32),
                          np.ascontiguousarray(b, np.float32).view(np.uint32))
