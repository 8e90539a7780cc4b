"""The N>1 path of bench.py (sharding + pose gather) on CPU with gloo, world_size 2.

The frame pairs are independent, so the only collective is the all-gather of the resulting poses
(SURVEY.md 8(e)).  Solving needs a GPU; here the poses are rank-stamped stand-ins so that the
partitioning and the gather ordering are what is tested."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from unified_cvo_amd import sharding


def test_partition_is_contiguous_and_complete():
    for total, world in ((512, 8), (10, 3), (5, 8), (64, 1)):
        seen = []
        for r in range(world):
            lo, hi = sharding.shard_range(total, world, r)
            assert 0 <= lo <= hi <= total
            seen += list(range(lo, hi))
        assert seen == list(range(total))
    assert sharding.shard_range(512, 8, 3) == (192, 256)  # pair p -> GPU p / 64


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sharding.shard_range(total, world, rank)
    local = torch.zeros(hi - lo, 16)
    for i, p in enumerate(range(lo, hi)):
        local[i] = torch.arange(16, dtype=torch.float32) + 100.0 * p  # stand-in for pair p's 4x4
    status = torch.full((hi - lo,), rank, dtype=torch.int32)
    poses, stat = sharding.gather_poses(local, status, total, world, rank)
    t = sharding.max_over_ranks(float(rank + 1))
    dist.barrier()
    if rank == 0:
        q.put((poses.numpy(), stat.numpy(), t))
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 7])
def test_gather_poses_world2(total):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    poses, stat, t = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert poses.shape == (total, 16)
    for p in range(total):
        assert np.array_equal(poses[p], np.arange(16, dtype=np.float32) + 100.0 * p)
    lo1, _ = sharding.shard_range(total, 2, 1)
    assert (stat[:lo1] == 0).all() and (stat[lo1:] == 1).all()
    assert t == 2.0
