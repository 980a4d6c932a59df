"""Host-side mirror of the reference's operator API for the compute path.

Names and argument meaning follow src/renderer/mod.rs (`Renderer`, `Settings`, `SaveInfo`),
src/renderer/scene.rs (`Scene`) and src/lib.rs (`load_ron`); all arithmetic happens in libmi355pt.so.
"""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import binding
from .binding import Mi355ptError, lib

SPHERE_DT = np.dtype([("pos", "<f4", 3), ("rad", "<f4"), ("mat_type", "<u4"), ("mat_index", "<u4"),
                      ("pad", "<u4", 2)])      # sphere.rs:26-43
VEC16_DT = np.dtype([("v", "<f4", 3), ("w", "<f4")])  # materials.rs Bytes impls (w = roughness for Metal)
NODE_DT = np.dtype([("mn", "<f4", 3), ("pad0", "<u4"), ("mx", "<f4", 3), ("pad1", "<u4"), ("type", "<u4"),
                    ("index", "<u4"), ("pad2", "<u4", 2)])  # bvh.rs:20-36


def _u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


@dataclass
class CameraSettings:  # camera.rs:3-22
    pos: tuple = (0.0, 0.0, -3.0)
    yaw: float = 0.0
    pitch: float = 0.0
    vfov: float = 60.0


@dataclass
class Settings:  # mod.rs:26-30
    samples: int = 1
    depth: int = 50


@dataclass
class Scene:  # scene.rs:5-14, arrays already in the reference's upload format
    camera: CameraSettings = field(default_factory=CameraSettings)
    spheres: np.ndarray = None
    lights: np.ndarray = None
    lambertians: np.ndarray = None
    metals: np.ndarray = None
    glass: np.ndarray = None

    @staticmethod
    def _from_handle(h):
        L = lib()
        out = Scene()
        dts = [SPHERE_DT, VEC16_DT, VEC16_DT, VEC16_DT, np.dtype("<f4")]
        arrays = []
        for which, dt in enumerate(dts):
            p = C.c_void_p()
            n = L.mi355pt_scene_array(h, which, C.byref(p))
            if n:
                buf = (C.c_char * (n * dt.itemsize)).from_address(p.value)
                arrays.append(np.frombuffer(buf, dtype=dt).copy())
            else:
                arrays.append(np.zeros(0, dt))
        out.spheres, out.lights, out.lambertians, out.metals, out.glass = arrays
        cam = (C.c_float * 6)()
        L.mi355pt_scene_camera(h, cam)
        out.camera = CameraSettings(tuple(np.float32(x) for x in cam[:3]), np.float32(cam[3]), np.float32(cam[4]),
                                    np.float32(cam[5]))
        return out

    @staticmethod
    def default():
        """Scene::default() (scene.rs:15-31)."""
        L = lib()
        h = L.mi355pt_scene_default()
        try:
            return Scene._from_handle(h)
        finally:
            L.mi355pt_scene_free(h)

    @staticmethod
    def from_ron_text(text):
        L = lib()
        h = C.c_void_p()
        err = C.create_string_buffer(512)
        data = text.encode() if isinstance(text, str) else text
        r = L.mi355pt_scene_parse_ron(data, len(data), C.byref(h), err, len(err))
        if r != binding.OK:
            raise Mi355ptError(r, err.value.decode(errors="replace"))
        try:
            return Scene._from_handle(h)
        finally:
            L.mi355pt_scene_free(h)


def load_ron_scene(path):
    """load_ron::<Scene>(path) (lib.rs:6-14): returns None on any IO / parse error."""
    L = lib()
    h = C.c_void_p()
    err = C.create_string_buffer(512)
    r = L.mi355pt_scene_load_ron(str(path).encode(), C.byref(h), err, len(err))
    if r != binding.OK:
        return None
    try:
        return Scene._from_handle(h)
    finally:
        L.mi355pt_scene_free(h)


def load_ron_settings(path):
    """load_ron::<Settings>(path): returns None on any IO / parse error."""
    s, d = C.c_int32(), C.c_int32()
    err = C.create_string_buffer(512)
    r = lib().mi355pt_settings_load_ron(str(path).encode(), C.byref(s), C.byref(d), err, len(err))
    return Settings(s.value, d.value) if r == binding.OK else None


def camera_build(pos, yaw_deg, pitch_deg, vfov_deg, aspect):
    """CameraConfig::new(settings, aspect).build().bytes() as 16 float32 (camera.rs:36-57,71-113)."""
    out = np.zeros(16, np.float32)
    p = (C.c_float * 3)(*[float(x) for x in pos])
    r = lib().mi355pt_camera_build(p, float(yaw_deg), float(pitch_deg), float(vfov_deg), float(aspect),
                                   out.ctypes.data)
    if r != binding.OK:
        raise Mi355ptError(r, "camera_build")
    return out


def bvh_build_reference(spheres, swap_mode=0, swap_seed=1):
    """flatten(BVHTree::new(spheres)) as NODE_DT records (bvh.rs:51-140)."""
    spheres = np.ascontiguousarray(spheres, SPHERE_DT)
    n = len(spheres)
    nodes = np.zeros(max(2 * n - 1, 1), NODE_DT)
    r = lib().mi355pt_bvh_build_reference(spheres.ctypes.data, n, swap_mode, swap_seed, nodes.ctypes.data, len(nodes))
    if r < 0:
        raise Mi355ptError(int(r), lib().mi355pt_last_error(None).decode(errors="replace"))
    return nodes[:r]


def partition_frames(total, parts, part):
    """(first, count) of the frames part `part` of `parts` renders: mi355pt_partition_frames."""
    first, count = C.c_uint32(0), C.c_uint32(0)
    lib().mi355pt_partition_frames(int(total), int(parts), int(part), C.byref(first), C.byref(count))
    return first.value, count.value


def reduce(renderers, root=0):
    """mi355pt_reduce: sums the accumulators of several Renderers (one process, any devices) into renderers[root]."""
    arr = (C.c_void_p * len(renderers))(*[r._ctx for r in renderers])
    rc = lib().mi355pt_reduce(arr, len(renderers), int(root))
    if rc != 0:
        # failures that name no context (bad root, empty list) are kept per thread, not on a context
        ctx = renderers[root]._ctx if 0 <= int(root) < len(renderers) else None
        msg = lib().mi355pt_last_error(ctx) if ctx else b""
        raise Mi355ptError(rc, (msg or lib().mi355pt_last_error(None) or b"").decode(errors="replace"))


def frame_seed(base_seed, frame):
    return int(lib().mi355pt_frame_seed(base_seed & 0xFFFFFFFF, frame & 0xFFFFFFFF))


def postprocess_rgb8(accum, frames):
    """SaveInfo::finish pixel math (mod.rs:420-432) on an [H,W,4] float32 array."""
    a = np.ascontiguousarray(accum, np.float32)
    h, w = a.shape[:2]
    out = np.zeros((h, w, 3), np.uint8)
    r = lib().mi355pt_postprocess_rgb8(_f32p(a), w, h, int(frames), _u8p(out))
    if r != binding.OK:
        raise Mi355ptError(r, "postprocess_rgb8")
    return out


def write_png_rgb8(path, rgb):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w = rgb.shape[:2]
    r = lib().mi355pt_write_png_rgb8(str(path).encode(), _u8p(rgb), w, h)
    if r != binding.OK:
        raise Mi355ptError(r, lib().mi355pt_last_error(None).decode(errors="replace"))


class SaveInfo:
    """What Renderer::start_save returns (mod.rs:388-444): finish() writes the PNG."""

    def __init__(self, renderer):
        self._r = renderer
        self.tex_width, self.tex_height = renderer.width, renderer.height

    def finish(self, samples, path="img.png"):
        self._r._check(lib().mi355pt_save_png(self._r._ctx, int(samples), str(path).encode()))


class Renderer:
    """Renderer (mod.rs:91-320) on one MI355X.

    render(n) accumulates n frames (n iterations of the reference's render + submit + poll loop,
    offline.rs:44-52); everything else keeps the reference's verbs.
    """

    def __init__(self, scene, settings, width, height, device=0, base_seed=0x00C0FFEE, bvh_mode=binding.BVH_SAH,
                 flags=0, stream=None, bvh_nodes=None):
        L = lib()
        self._ctx = C.c_void_p()
        self.width, self.height = int(width), int(height)
        desc = binding.Desc(self.width, self.height, int(device), base_seed & 0xFFFFFFFF, int(bvh_mode), int(flags),
                            C.c_void_p(stream) if stream else None)
        r = L.mi355pt_create(C.byref(desc), C.byref(self._ctx))
        if r != binding.OK:
            self._ctx = None
            raise Mi355ptError(r, L.mi355pt_last_error(None).decode(errors="replace"))
        self.reload_settings(settings)
        self.reload_scene(scene, bvh_nodes=bvh_nodes)

    # -- plumbing
    def _check(self, r):
        if r != binding.OK:
            raise Mi355ptError(r, lib().mi355pt_last_error(self._ctx).decode(errors="replace"))

    def close(self):
        if getattr(self, "_ctx", None):
            lib().mi355pt_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- reference verbs
    def reload_scene(self, scene, bvh_nodes=None):  # mod.rs:148-155
        sp = np.ascontiguousarray(scene.spheres, SPHERE_DT)
        li = np.ascontiguousarray(scene.lights, VEC16_DT)
        la = np.ascontiguousarray(scene.lambertians, VEC16_DT)
        me = np.ascontiguousarray(scene.metals, VEC16_DT)
        gl = np.ascontiguousarray(scene.glass, np.float32)
        L = lib()
        self._check(L.mi355pt_set_scene(self._ctx, sp.ctypes.data, len(sp), li.ctypes.data, len(li), la.ctypes.data,
                                        len(la), me.ctypes.data, len(me), gl.ctypes.data, len(gl)))
        if bvh_nodes is not None:
            nodes = np.ascontiguousarray(bvh_nodes, NODE_DT)
            self._check(L.mi355pt_set_bvh(self._ctx, nodes.ctypes.data, len(nodes)))
        cam = scene.camera
        pos = (C.c_float * 3)(*[float(x) for x in cam.pos])
        self._check(L.mi355pt_set_camera_config(self._ctx, pos, float(cam.yaw), float(cam.pitch), float(cam.vfov)))
        self._camera = cam

    def reload_settings(self, settings):  # mod.rs:157-160
        self._check(lib().mi355pt_set_settings(self._ctx, int(settings.samples), int(settings.depth)))
        self.settings = settings

    def resize(self, width, height):  # mod.rs:162-166
        self._check(lib().mi355pt_resize(self._ctx, int(width), int(height)))
        self.width, self.height = int(width), int(height)

    def render(self, n_frames=1, seeds=None, wait=True):  # mod.rs:168-244, n_frames times
        """wait=False queues the frames and returns (mi355pt_render_async); `wait()` or any reading call
        joins."""
        if seeds is not None:
            seeds = np.ascontiguousarray(seeds, np.uint32)
            n_frames = len(seeds)
            p = seeds.ctypes.data_as(C.POINTER(C.c_uint32))
        else:
            p = None
        fn = lib().mi355pt_render if wait else lib().mi355pt_render_async
        self._check(fn(self._ctx, int(n_frames), p))

    def render_static(self, n_frames=1, seeds=None):
        """mi355pt_render_static: the same frames without any device allocation (static workspace)."""
        if seeds is not None:
            seeds = np.ascontiguousarray(seeds, np.uint32)
            n_frames = len(seeds)
            p = seeds.ctypes.data_as(C.POINTER(C.c_uint32))
        else:
            p = None
        self._check(lib().mi355pt_render_static(self._ctx, int(n_frames), p))

    def wait(self):
        self._check(lib().mi355pt_wait(self._ctx))

    def samples(self):  # mod.rs:246-248
        return int(lib().mi355pt_frames(self._ctx))

    def camera_config(self):  # mod.rs:259-261
        return self._camera

    def set_camera_config(self, cam):  # camera_config_mut (mod.rs:254-257): marks dirty
        pos = (C.c_float * 3)(*[float(x) for x in cam.pos])
        self._check(lib().mi355pt_set_camera_config(self._ctx, pos, float(cam.yaw), float(cam.pitch),
                                                    float(cam.vfov)))
        self._camera = cam

    def set_camera_bytes(self, cam64):
        cam64 = np.ascontiguousarray(cam64, np.float32)
        assert cam64.nbytes == 64
        self._check(lib().mi355pt_set_camera(self._ctx, cam64.ctypes.data))

    def camera_bytes(self):
        out = np.zeros(16, np.float32)
        self._check(lib().mi355pt_get_camera(self._ctx, out.ctypes.data))
        return out

    def clear(self):
        self._check(lib().mi355pt_clear(self._ctx))

    def start_save(self):  # mod.rs:263-315
        return SaveInfo(self)

    # -- outputs beyond the reference's API
    def read_accum(self):
        """The RGBA32F accumulation texture as [H, W, 4] float32."""
        out = np.empty((self.height, self.width, 4), np.float32)
        self._check(lib().mi355pt_read_accum(self._ctx, _f32p(out), out.size))
        return out

    def write_accum(self, accum, frames):
        a = np.ascontiguousarray(accum, np.float32)
        self._check(lib().mi355pt_write_accum(self._ctx, _f32p(a), a.size, int(frames)))

    def to_rgb8(self, frames=None):
        out = np.empty((self.height, self.width, 3), np.uint8)
        frames = self.samples() if frames is None else frames
        self._check(lib().mi355pt_to_rgb8(self._ctx, int(frames), _u8p(out), out.size))
        return out

    def stats(self):
        st = binding.Stats()
        self._check(lib().mi355pt_stats_get(self._ctx, C.byref(st)))
        d = {name: getattr(st, name) for name, _ in binding.Stats._fields_ if name not in ("bounces", "phase", "exchange")}
        d["exchange"] = dict(zip(["deposits", "lanes_deposited", "withdrawals", "lanes_withdrawn"], [int(x) for x in st.exchange]))
        d["bounces"] = dict(zip(["light", "lambertian", "metal", "glass", "sky"], [int(x) for x in st.bounces]))
        ph = [int(x) for x in st.phase]
        d["phases"] = {k: {"executions": ph[2 * i], "lanes": ph[2 * i + 1]}
                       for i, k in enumerate(["node", "leaf", "terminal", "opaque", "glass", "idle"])}
        return d

    def accum_device_ptr(self):
        """(device pointer of the float4[W*H] accumulator, hipStream_t it is written on)."""
        p, s = C.c_void_p(), C.c_void_p()
        self._check(lib().mi355pt_accum_device_ptr(self._ctx, C.byref(p), C.byref(s)))
        return p.value, s.value
