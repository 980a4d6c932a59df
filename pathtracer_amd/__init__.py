"""pathtracer_amd -- Python mirror of the reference's `Renderer` API over libmi355pt.so (HIP, gfx950).

The compute path lives in the shared library (pathtracer_amd/csrc, C ABI in include/mi355pt.h); this
package only binds it with ctypes.  There is no CPU path: constructing a Renderer without a usable
HIP device raises.
"""
from .binding import (BVH_LBVH, BVH_REFERENCE, BVH_SAH, FLAG_COUNT, FLAG_GLASS_SHORT_CIRCUIT, Mi355ptError, lib,
                      library_path)
from .renderer import (CameraSettings, Renderer, SaveInfo, Scene, Settings, bvh_build_reference, camera_build,
                       frame_seed, load_ron_scene, load_ron_settings, partition_frames, postprocess_rgb8, reduce,
                       write_png_rgb8)

__all__ = [
    "BVH_LBVH", "BVH_REFERENCE", "BVH_SAH", "FLAG_COUNT", "FLAG_GLASS_SHORT_CIRCUIT", "Mi355ptError", "lib", "library_path",
    "CameraSettings", "Renderer", "SaveInfo", "Scene", "Settings", "bvh_build_reference", "camera_build",
    "frame_seed", "load_ron_scene", "load_ron_settings", "partition_frames", "postprocess_rgb8", "reduce", "write_png_rgb8",
]
