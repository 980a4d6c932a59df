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
    L.emul_set_brute_force.argtypes = [C.c_int]
    L.emul_set_tile_lists.argtypes = [C.c_int, C.c_uint32]

    def tile_list_stats():
        out = (C.c_uint64 * 4)()
        L.emul_tile_list_stats(out)
        return dict(zip(["list_rays", "list_tests", "walk_tiles", "list_tiles"], [int(v) for v in out]))

    L.tile_list_stats = tile_list_stats

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

    def tile_lists(sc, cam, W, H, max_list=15):
        """(lists[tiles_y, tiles_x, 16], order[n_spheres]): word 0 = count or 0xffffffff (keep the walk), then
        leaf-order sphere indices; order maps those to the scene's own sphere indices."""
        ty, tx = (H + 7) // 8, (W + 7) // 8
        out = np.zeros((ty, tx, 16), np.uint32)
        order = np.zeros(len(sc.spheres), np.uint32)
        cam = np.ascontiguousarray(cam, np.float32)
        vp, sz = C.c_void_p, C.c_size_t
        r = L.emul_tile_lists(vp(sc.spheres.ctypes.data), sz(len(sc.spheres)), vp(sc.lights.ctypes.data),
                              sz(len(sc.lights)), vp(sc.lambertians.ctypes.data), sz(len(sc.lambertians)),
                              vp(sc.metals.ctypes.data), sz(len(sc.metals)), vp(sc.glass.ctypes.data), sz(len(sc.glass)),
                              vp(cam.ctypes.data), C.c_uint32(W), C.c_uint32(H), C.c_uint32(max_list),
                              vp(out.ctypes.data), vp(order.ctypes.data))
        if r != 0:
            raise RuntimeError(L.emul_error().decode())
        return out, order

    L.tile_lists = tile_lists
    L.emul_set_blob.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]

    def set_blob(blob=None, layout=None):
        """Walk this packed scene (device_scene_blob) in render() instead of the harness's own packing; None: back."""
        if blob is None:
            return L.emul_set_blob(None, 0, None, 0)
        r = L.emul_set_blob(blob.ctypes.data, blob.nbytes, layout.ctypes.data, layout.nbytes)
        if r != 0:
            raise RuntimeError("emul_set_blob: layout and blob do not match")

    L.set_blob = set_blob
    return L


SCENE_LAYOUT_BYTES = 84   # sizeof(pt::SceneLayout) (pt_types.h); the library and the harness both check it


def device_scene_blob(renderer):
    """(blob, layout) of the packed scene a Renderer's device holds, tree included, whoever built it: the library's
    test-only export mi355pt_debug_scene_blob."""
    import ctypes as C
    from pathtracer_amd import binding
    L = C.CDLL(binding.library_path())
    fn = L.mi355pt_debug_scene_blob
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t]
    layout = np.zeros(SCENE_LAYOUT_BYTES, np.uint8)
    size = C.c_size_t(0)
    assert fn(renderer._ctx, None, 0, C.byref(size), layout.ctypes.data, layout.nbytes) == 0
    blob = np.zeros(size.value, np.uint8)
    assert fn(renderer._ctx, blob.ctypes.data, blob.nbytes, C.byref(size), layout.ctypes.data, layout.nbytes) == 0
    return blob, layout


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


def random_scene(pa, seed, duplicates=True):
    """A small random scene that leans on the awkward cases: overlapping and nested spheres, the camera
    possibly inside one, exact duplicates (equal roots), glass with ir below, at and above 1, mirror and
    fully rough metals, bright lights, now and then a material type the shader does not know."""
    from pathtracer_amd.renderer import SPHERE_DT, VEC16_DT, CameraSettings, Scene
    rng = np.random.default_rng(seed)
    n = int(rng.integers(2, 60))
    sp = np.zeros(n, SPHERE_DT)
    sp["pos"] = rng.uniform((-6, -1, -6), (6, 3, 6), (n, 3)).astype(np.float32)
    sp["rad"] = rng.choice([0.05, 0.3, 0.5, 1.0, 1.5, 4.0], n, p=[.1, .3, .3, .15, .1, .05]).astype(np.float32)
    if rng.random() < 0.7:                                   # a ground sphere, as in every shipped scene
        sp[0] = ((0.0, -1000.0, 0.0), 1000.0, 1, 0, (0, 0))
    counts = rng.integers(1, 6, 4)
    sp["mat_type"] = rng.integers(0, 4, n)
    sp["mat_index"] = [rng.integers(0, counts[t]) for t in sp["mat_type"]]
    unknown = rng.random(n) < 0.03
    sp["mat_type"][unknown] = 9                              # compute.wgsl:356-358 -> black
    if duplicates and n >= 4:
        k = int(rng.integers(1, 4))
        src, dst = rng.integers(0, n, k), rng.integers(1, n, k)
        for a, b in zip(src, dst):                           # same ball, usually another material
            sp["pos"][b], sp["rad"][b] = sp["pos"][a], sp["rad"][a]
    lights = np.zeros(counts[0], VEC16_DT)
    lights["v"] = rng.uniform(0.5, 6.0, (counts[0], 3)).astype(np.float32)
    lam = np.zeros(counts[1], VEC16_DT)
    lam["v"] = rng.uniform(0.0, 1.0, (counts[1], 3)).astype(np.float32)
    met = np.zeros(counts[2], VEC16_DT)
    met["v"] = rng.uniform(0.2, 1.0, (counts[2], 3)).astype(np.float32)
    met["w"] = rng.choice([0.0, 0.1, 0.5, 1.0], counts[2]).astype(np.float32)
    gla = rng.choice([0.7, 1.0, 1.01, 1.5, 2.4], counts[3]).astype(np.float32)
    cam = CameraSettings(tuple(rng.uniform((-8, 0, -8), (8, 4, 8)).astype(np.float32)),
                         np.float32(rng.uniform(-180, 180)), np.float32(rng.uniform(-40, 20)),
                         np.float32(rng.choice([20.0, 45.0, 70.0, 90.0])))
    return Scene(cam, sp, lights, lam, met, gla)
