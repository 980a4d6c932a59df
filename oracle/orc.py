"""ctypes binding of the CPU oracle (oracle/libpt_oracle.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

from . import ronlite  # noqa: F401 (re-exported as orc.ronlite)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libpt_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "pt_oracle.cpp")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libpt_oracle.so"])
    return _LIB


class OrcScene(C.Structure):
    _fields_ = [("bvh", C.c_void_p), ("n_nodes", C.c_uint32),
                ("spheres", C.c_void_p), ("n_spheres", C.c_uint32),
                ("lights", C.c_void_p), ("n_lights", C.c_uint32),
                ("lambertians", C.c_void_p), ("n_lambertians", C.c_uint32),
                ("metals", C.c_void_p), ("n_metals", C.c_uint32),
                ("glass", C.c_void_p), ("n_glass", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        f32p = C.POINTER(C.c_float)
        u32p = C.POINTER(C.c_uint32)
        L.orc_randu.restype = C.c_uint32
        L.orc_randu.argtypes = [u32p]
        L.orc_randf.restype = C.c_float
        L.orc_randf.argtypes = [u32p]
        L.orc_rand_in_sphere.argtypes = [u32p, f32p]
        L.orc_det_sin.restype = C.c_float
        L.orc_det_sin.argtypes = [C.c_float]
        L.orc_vec2rand.restype = C.c_float
        L.orc_vec2rand.argtypes = [C.c_float, C.c_float]
        L.orc_sat_u32.restype = C.c_uint32
        L.orc_sat_u32.argtypes = [C.c_float]
        L.orc_pixel_seed.restype = C.c_uint32
        L.orc_pixel_seed.argtypes = [C.c_uint32] * 5
        L.orc_miss.argtypes = [C.c_float, f32p]
        L.orc_reflectance.restype = C.c_float
        L.orc_reflectance.argtypes = [C.c_float, C.c_float]
        L.orc_refract.argtypes = [f32p, f32p, C.c_float, f32p]
        L.orc_reflect.argtypes = [f32p, f32p, f32p]
        L.orc_normalize.argtypes = [f32p, f32p]
        L.orc_ray_sphere.restype = C.c_int
        L.orc_ray_sphere.argtypes = [C.c_void_p, f32p, f32p, C.c_float, C.c_float, f32p]
        L.orc_ray_bbox.restype = C.c_int
        L.orc_ray_bbox.argtypes = [f32p, f32p, f32p, f32p]
        L.orc_vector3.argtypes = [C.c_int, f32p, f32p, f32p]
        L.orc_camera_build.argtypes = [f32p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]
        L.orc_bvh_build.restype = C.c_int64
        L.orc_bvh_build.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, C.c_void_p, C.c_uint64]
        L.orc_closest_hit.restype = C.c_int
        L.orc_closest_hit.argtypes = [C.POINTER(OrcScene), f32p, f32p, f32p, u32p, C.POINTER(C.c_uint64)]
        L.orc_camera_ray.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_uint32,
                                     C.c_uint32, f32p]
        L.orc_render.argtypes = [C.POINTER(OrcScene), C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32,
                                 C.c_int32, C.c_uint32, u32p, f32p, C.c_uint32, C.c_uint32, C.c_uint32,
                                 C.c_uint32, C.c_int, C.POINTER(C.c_uint64)]
        L.orc_shade_pixel.argtypes = [C.POINTER(OrcScene), C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32,
                                      C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32, f32p]
        L.orc_postprocess.argtypes = [f32p, C.c_uint32, C.c_uint32, C.c_int32, C.POINTER(C.c_uint8)]
        L.orc_set_glass_always_draws.argtypes = [C.c_int]
        _lib = L
    return _lib


def _f3(a):
    return (C.c_float * 3)(*[float(x) for x in a])


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


STAT_NAMES = ["paths", "rays", "interior_visits", "leaf_visits", "accepted", "bounce_light",
              "bounce_lambertian", "bounce_metal", "bounce_glass", "misses", "exhausted"]


def camera_build(pos, yaw_deg, pitch_deg, vfov_deg, aspect):
    out = np.zeros(16, np.float32)
    lib().orc_camera_build(_f3(pos), float(yaw_deg), float(pitch_deg), float(vfov_deg), float(aspect),
                           out.ctypes.data)
    return out


def bvh_build(spheres, swap_mode=0, swap_seed=1):
    n = len(spheres)
    nodes = np.zeros(max(2 * n - 1, 1), ronlite.NODE_DT)
    r = lib().orc_bvh_build(spheres.ctypes.data, n, swap_mode, swap_seed, nodes.ctypes.data, len(nodes))
    if r < 0:
        raise ValueError(f"orc_bvh_build failed: {r}")
    return nodes[:r]


class Scene:
    """Byte blobs of one scene + its reference-topology BVH, ready for the oracle."""

    def __init__(self, blobs, swap_mode=0, swap_seed=1, nodes=None):
        self.blobs = blobs
        self.spheres = np.ascontiguousarray(blobs["spheres"])
        self.lights = np.ascontiguousarray(blobs["lights"])
        self.lambertians = np.ascontiguousarray(blobs["lambertians"])
        self.metals = np.ascontiguousarray(blobs["metals"])
        self.glass = np.ascontiguousarray(blobs["glass"])
        self.nodes = bvh_build(self.spheres, swap_mode, swap_seed) if nodes is None else nodes
        s = OrcScene()
        s.bvh, s.n_nodes = self.nodes.ctypes.data, len(self.nodes)
        s.spheres, s.n_spheres = self.spheres.ctypes.data, len(self.spheres)
        s.lights, s.n_lights = self.lights.ctypes.data, len(self.lights)
        s.lambertians, s.n_lambertians = self.lambertians.ctypes.data, len(self.lambertians)
        s.metals, s.n_metals = self.metals.ctypes.data, len(self.metals)
        s.glass, s.n_glass = self.glass.ctypes.data, len(self.glass)
        self.c = s

    @classmethod
    def from_ron(cls, path, **kw):
        return cls(ronlite.scene_blobs(ronlite.load(path)), **kw)

    def camera(self, width, height):
        pos, yaw, pitch, vfov = self.blobs["camera"]
        return camera_build(pos, yaw, pitch, vfov, np.float32(width) / np.float32(height))


def render(scene, cam64, width, height, seeds, samples=1, depth=50, accum=None, crop=None, threads=1):
    """Accumulate len(seeds) frames.  Returns (accum[H,W,4] float32, stats dict)."""
    seeds = np.ascontiguousarray(seeds, np.uint32)
    if accum is None:
        accum = np.zeros((height, width, 4), np.float32)
        accum[..., 3] = 1.0
    x0, x1, y0, y1 = crop if crop else (0, width, 0, height)
    st = (C.c_uint64 * 16)()
    cam64 = np.ascontiguousarray(cam64, np.float32)
    lib().orc_render(C.byref(scene.c), cam64.ctypes.data, width, height, samples, depth, len(seeds),
                     seeds.ctypes.data_as(C.POINTER(C.c_uint32)), _fp(accum), x0, x1, y0, y1, threads, st)
    return accum, dict(zip(STAT_NAMES, [int(x) for x in st[:11]]))


def shade_pixel(scene, cam64, width, height, seed, px, py, samples=1, depth=50):
    out = (C.c_float * 3)()
    cam64 = np.ascontiguousarray(cam64, np.float32)
    lib().orc_shade_pixel(C.byref(scene.c), cam64.ctypes.data, width, height, samples, depth, seed, px, py,
                          out)
    return np.array(out[:], np.float32)


def closest_hit(scene, o, d):
    rec = (C.c_float * 8)()
    idx = C.c_uint32(0)
    vis = (C.c_uint64 * 3)()
    ok = lib().orc_closest_hit(C.byref(scene.c), _f3(o), _f3(d), rec, C.byref(idx), vis)
    return bool(ok), np.array(rec[:], np.float32), idx.value, [int(v) for v in vis]


def camera_ray(cam64, px, py, jx, jy, width, height):
    out = (C.c_float * 6)()
    cam64 = np.ascontiguousarray(cam64, np.float32)
    lib().orc_camera_ray(cam64.ctypes.data, px, py, float(jx), float(jy), width, height, out)
    return np.array(out[:3], np.float32), np.array(out[3:], np.float32)


def postprocess(accum, frames):
    h, w = accum.shape[:2]
    out = np.zeros((h, w, 3), np.uint8)
    a = np.ascontiguousarray(accum, np.float32)
    lib().orc_postprocess(_fp(a), w, h, frames, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out
