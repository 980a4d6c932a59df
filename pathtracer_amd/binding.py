"""ctypes declarations for every entry point of include/mi355pt.h."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MI355PT_LIB: load another build of the same library (A/B measurements of two builds)
_LIB_PATH = os.environ.get("MI355PT_LIB") or os.path.join(_HERE, "libmi355pt.so")

BVH_SAH = 0
BVH_REFERENCE = 1
BVH_LBVH = 2
FLAG_GLASS_SHORT_CIRCUIT = 1 << 0
FLAG_COUNT = 1 << 1

OK = 0


class Mi355ptError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"mi355pt error {code}: {message}")
        self.code = code


class Desc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("device", C.c_int32),
                ("base_seed", C.c_uint32), ("bvh_mode", C.c_uint32), ("flags", C.c_uint32),
                ("stream", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays", C.c_uint64), ("nodes", C.c_uint64), ("tests", C.c_uint64),
                ("accepted", C.c_uint64), ("bounces", C.c_uint64 * 5), ("phase", C.c_uint64 * 12),
                ("render_ms", C.c_double),
                ("resolve_ms", C.c_double), ("launches", C.c_uint32), ("bvh_nodes", C.c_uint32),
                ("bvh_depth", C.c_uint32), ("scene_bytes", C.c_uint32), ("scene_in_lds", C.c_uint32),
                ("node_format", C.c_uint32), ("exchange_on", C.c_uint32), ("exchange", C.c_uint64 * 4),
                ("list_tiles", C.c_uint32), ("list_candidates", C.c_uint32),
                ("sky_tiles", C.c_uint32), ("short_div", C.c_uint32)]


# name -> (restype, argtypes); the single list the symbol-export test walks
SIGNATURES = {
    "mi355pt_create": (C.c_int, [C.POINTER(Desc), C.POINTER(C.c_void_p)]),
    "mi355pt_destroy": (None, [C.c_void_p]),
    "mi355pt_last_error": (C.c_char_p, [C.c_void_p]),
    "mi355pt_set_scene": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                    C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "mi355pt_set_bvh": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "mi355pt_set_camera": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mi355pt_set_camera_config": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float]),
    "mi355pt_get_camera": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mi355pt_set_settings": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "mi355pt_resize": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "mi355pt_clear": (C.c_int, [C.c_void_p]),
    "mi355pt_render": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]),
    "mi355pt_render_async": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]),
    "mi355pt_render_static": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]),
    "mi355pt_wait": (C.c_int, [C.c_void_p]),
    "mi355pt_partition_frames": (None, [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "mi355pt_reduce": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t]),
    "mi355pt_reduce_comm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int32]),
    "mi355pt_frames": (C.c_int32, [C.c_void_p]),
    "mi355pt_read_accum": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_size_t]),
    "mi355pt_write_accum": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_size_t, C.c_int32]),
    "mi355pt_to_rgb8": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_uint8), C.c_size_t]),
    "mi355pt_save_png": (C.c_int, [C.c_void_p, C.c_int32, C.c_char_p]),
    "mi355pt_stats_get": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "mi355pt_accum_device_ptr": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "mi355pt_camera_build": (C.c_int, [C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float, C.c_float,
                                       C.c_void_p]),
    "mi355pt_bvh_build_reference": (C.c_int64, [C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_void_p,
                                                C.c_size_t]),
    "mi355pt_frame_seed": (C.c_uint32, [C.c_uint32, C.c_uint32]),
    "mi355pt_postprocess_rgb8": (C.c_int, [C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.c_int32,
                                           C.POINTER(C.c_uint8)]),
    "mi355pt_write_png_rgb8": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32]),
    "mi355pt_scene_load_ron": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]),
    "mi355pt_scene_parse_ron": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]),
    "mi355pt_scene_default": (C.c_void_p, []),
    "mi355pt_scene_free": (None, [C.c_void_p]),
    "mi355pt_scene_array": (C.c_size_t, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "mi355pt_scene_camera": (None, [C.c_void_p, C.POINTER(C.c_float)]),
    "mi355pt_set_scene_handle": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mi355pt_settings_load_ron": (C.c_int, [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_char_p,
                                            C.c_size_t]),
}

_lib = None


def library_path():
    return _LIB_PATH


def lib():
    """Loads libmi355pt.so (built in-tree by `make -C pathtracer_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise ImportError(f"{_LIB_PATH} is missing: build it with `make -C pathtracer_amd/csrc` "
                              "(there is no fallback implementation)")
        L = C.CDLL(_LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib
