"""Batched multi-GPU mode: independent frame pairs are sharded across ranks (one process per GPU) and
the resulting poses are all-gathered once per batch (RCCL over xGMI with backend "nccl"; gloo on CPU).

There is no collective inside the optimiser loop: every align() is independent."""
import torch
import torch.distributed as dist


def shard_range(total, world, rank):
    """Contiguous block partition; for 512 pairs over 8 ranks pair p lives on rank p // 64."""
    per = (total + world - 1) // world
    lo = min(rank * per, total)
    return lo, min(lo + per, total)


def gather_poses(local_poses, local_status, total, world, rank):
    """All-gathers (n_local x 16) poses + per-pair status into (total x 16), (total,) in pair order.

    Uneven shards are padded to the largest shard so one all_gather_into_tensor suffices."""
    per = (total + world - 1) // world
    dev = local_poses.device
    pad_p = torch.zeros(per, 16, dtype=local_poses.dtype, device=dev)
    pad_s = torch.zeros(per, dtype=local_status.dtype, device=dev)
    n = local_poses.shape[0]
    pad_p[:n] = local_poses
    pad_s[:n] = local_status
    if not dist.is_initialized():
        return pad_p[:total], pad_s[:total]
    all_p = torch.empty(world * per, 16, dtype=local_poses.dtype, device=dev)
    all_s = torch.empty(world * per, dtype=local_status.dtype, device=dev)
    dist.all_gather_into_tensor(all_p, pad_p)
    dist.all_gather_into_tensor(all_s, pad_s)
    keep = []
    for r in range(world):
        lo, hi = shard_range(total, world, r)
        keep.append(torch.arange(r * per, r * per + (hi - lo), device=dev))
    idx = torch.cat(keep)
    return all_p[idx], all_s[idx]


def max_over_ranks(value, device="cpu"):
    """MAX-reduction of a python float over ranks (the bench's timing contract)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
