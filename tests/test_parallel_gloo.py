"""The N > 1 path (frame partition by rank + one sum-reduce to rank 0) on CPU with the gloo backend,
world_size 2 and 3.  Each rank's frames are rendered by the CPU oracle standing in for its GPU; the
partition, the reduce and the alpha fix-up are the library's own (pathtracer_amd/parallel.py)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import FIXTURES, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import pathtracer_amd as pa
    from oracle import orc
    from pathtracer_amd import parallel

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    osc = orc.Scene.from_ron(os.path.join(FIXTURES, "scene2.ron"))
    W, H = 48, 27
    frames = parallel.rank_frames(n_frames, rank, world)
    seeds = parallel.seeds_for(0x00C0FFEE, frames, pa.frame_seed)
    acc, _ = orc.render(osc, osc.camera(W, H), W, H, seeds)
    t = torch.from_numpy(acc)
    parallel.reduce_accum(t, world, dst=0)
    counts = torch.tensor([len(frames)], dtype=torch.int64)
    dist.all_reduce(counts)
    if rank == 0:
        np.save(out_path, np.concatenate([t.numpy().ravel(), [float(counts.item())]]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_frame_partition_and_reduce(world, tmp_path, orc, pa):
    import torch.multiprocessing as mp
    from pathtracer_amd import parallel
    n_frames = 7
    out = str(tmp_path / "acc.npy")
    mp.spawn(_worker, args=(world, _free_port(), n_frames, out), nprocs=world, join=True)
    got = np.load(out)
    assert got[-1] == n_frames
    got = got[:-1].astype(np.float32).reshape(27, 48, 4)
    osc = orc.Scene.from_ron(os.path.join(FIXTURES, "scene2.ron"))
    seeds = parallel.seeds_for(0x00C0FFEE, range(n_frames), pa.frame_seed)
    ref, _ = orc.render(osc, osc.camera(48, 27), 48, 27, seeds)
    # same sample set; only the order of the float additions across ranks differs
    assert np.all(got[..., 3] == 1.0)
    np.testing.assert_allclose(got[..., :3], ref[..., :3], rtol=2e-6, atol=1e-6)


def _step_worker(rank, world, port, steps, frames_per_step, total_frames, out_path):
    # bench.py's multi-step flow: every step renders its block of one seed schedule into the same accumulator,
    # one reduce at the end
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import pathtracer_amd as pa
    from oracle import orc
    from pathtracer_amd import parallel

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    osc = orc.Scene.from_ron(os.path.join(FIXTURES, "scene3.ron"))
    W, H = 40, 24
    acc, mine = None, 0
    for step in range(steps):
        frames = parallel.step_frames(step, frames_per_step, rank, world, total_frames)
        mine += len(frames)
        if len(frames):
            acc, _ = orc.render(osc, osc.camera(W, H), W, H, parallel.seeds_for(0x00C0FFEE, frames, pa.frame_seed), accum=acc)
    if acc is None:                       # a rank without frames still takes part in the reduce
        acc = np.zeros((H, W, 4), np.float32)
        acc[..., 3] = 1.0
    t = torch.from_numpy(acc)
    parallel.reduce_accum(t, world, dst=0)
    counts = torch.tensor([mine], dtype=torch.int64)
    dist.all_reduce(counts)
    if rank == 0:
        np.save(out_path, np.concatenate([t.numpy().ravel(), [float(counts.item())]]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,frames_per_step,total_frames", [(2, 2, 0), (3, 1, 0), (2, 0, 5), (3, 0, 5), (3, 0, 2)])
def test_bench_step_schedule_weak_and_strong(world, frames_per_step, total_frames, tmp_path, orc, pa):
    # both modes of bench.py (weak: frames_per_step per rank and step; strong: --total-frames per step over all
    # ranks, the shape of the eight-GPU configuration; with 2 frames over 3 ranks one rank has none) render exactly
    # the frames 0 .. steps * frames_all - 1 of one schedule, each once
    import torch.multiprocessing as mp
    from pathtracer_amd import parallel
    steps = 2
    frames_all = total_frames if total_frames else frames_per_step * world
    covered = sorted(f for s in range(steps) for r in range(world)
                     for f in parallel.step_frames(s, frames_per_step, r, world, total_frames))
    assert covered == list(range(steps * frames_all))
    out = str(tmp_path / "acc.npy")
    mp.spawn(_step_worker, args=(world, _free_port(), steps, frames_per_step, total_frames, out), nprocs=world, join=True)
    got = np.load(out)
    assert got[-1] == steps * frames_all
    got = got[:-1].astype(np.float32).reshape(24, 40, 4)
    osc = orc.Scene.from_ron(os.path.join(FIXTURES, "scene3.ron"))
    seeds = parallel.seeds_for(0x00C0FFEE, range(steps * frames_all), pa.frame_seed)
    ref, _ = orc.render(osc, osc.camera(40, 24), 40, 24, seeds)
    assert np.all(got[..., 3] == 1.0)
    np.testing.assert_allclose(got[..., :3], ref[..., :3], rtol=2e-6, atol=1e-6)


def test_rank_frames_partition_is_exact():
    from pathtracer_amd import parallel
    for n in (0, 1, 7, 64, 1000, 1024, 16384):
        for world in (1, 2, 3, 4, 8):
            parts = [parallel.rank_frames(n, r, world) for r in range(world)]
            flat = [f for p in parts for f in p]
            assert flat == list(range(n))                       # disjoint, complete, in order
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
