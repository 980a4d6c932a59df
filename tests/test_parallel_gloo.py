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


def test_rank_frames_partition_is_exact():
    from pathtracer_amd import parallel
    for n in (0, 1, 7, 64, 1000, 1024, 16384):
        for world in (1, 2, 3, 4, 8):
            parts = [parallel.rank_frames(n, r, world) for r in range(world)]
            flat = [f for p in parts for f in p]
            assert flat == list(range(n))                       # disjoint, complete, in order
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
