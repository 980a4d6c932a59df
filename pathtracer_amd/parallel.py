"""Sample-parallel rendering across the GPUs of one node: one process per GPU, frames partitioned by
index (disjoint seed streams), one sum-reduce of the accumulators to rank 0 at the end.

The reference is single-GPU; what makes this exact is that every (frame, pixel) path depends only on
the frame's seed and the pixel (compute.wgsl:400) and frames meet only in the per-texel sum
(compute.wgsl:419-421).  The union over ranks is the single-GPU sample set; only the order of the
float additions across ranks differs.
"""
import numpy as np


def rank_frames(n_frames, rank, world, first=0):
    """Global frame indices rendered by `rank` out of frames [first, first + n_frames): contiguous blocks,
    sizes differ by at most one -- the library's own split (mi355pt_partition_frames), which the offline
    CLI, bench.py and the tests all use."""
    from .renderer import partition_frames
    start, count = partition_frames(n_frames, world, rank)
    return range(first + start, first + start + count)


def step_frames(step, frames_per_step, rank, world, total_frames=0):
    """Frames `rank` renders in step `step` of a multi-step run (bench.py): every step covers the next
    `frames_all` frames of one seed schedule, split over the ranks by rank_frames.  Weak scaling (total_frames == 0):
    frames_all = frames_per_step * world, so every rank renders frames_per_step of them; strong scaling:
    frames_all = total_frames whatever the world size (the shape of the 8-GPU configuration)."""
    frames_all = total_frames if total_frames > 0 else frames_per_step * world
    return rank_frames(frames_all, rank, world, first=step * frames_all)


class _DevicePtr:
    """Minimal __cuda_array_interface__ carrier so torch can wrap memory the library allocated."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f4", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def accum_tensor(renderer, device_index):
    """The renderer's float4[H*W] accumulator as a torch tensor [H, W, 4] (no copy)."""
    import torch
    ptr, _stream = renderer.accum_device_ptr()
    return torch.as_tensor(_DevicePtr(ptr, (renderer.height, renderer.width, 4)),
                           device=torch.device("cuda", device_index))


def reduce_accum(tensor, world, dst=0):
    """Sum-reduce accumulators to `dst` (RCCL when the tensor is on a GPU, gloo on the CPU) and restore
    the constant alpha of the reference's texture (every rank cleared its own to 1)."""
    import torch.distributed as dist
    grouped = world > 1 or (dist.is_available() and dist.is_initialized())   # a group of one still reduces
    if grouped:
        if tensor.is_cuda and dist.get_backend() == "gloo":  # gloo reduces host tensors only
            host = tensor.cpu()
            dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM)
            if dist.get_rank() == dst:
                tensor.copy_(host)
        else:
            dist.reduce(tensor, dst=dst, op=dist.ReduceOp.SUM)
    if dist.get_rank() == dst if grouped else True:
        tensor[..., 3] = 1.0
    return tensor


def seeds_for(base_seed, frames, frame_seed_fn):
    return np.array([frame_seed_fn(base_seed, f) for f in frames], np.uint32)
