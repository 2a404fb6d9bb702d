"""Multi-GPU sharding of the MFCC path: one process per GPU (torch.distributed,
backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference has no distributed code at all.  Frames and clips
are independent, so the path shards embarrassingly: rank r owns one contiguous
block of items, computes it with its own plan on its own GPU, and the only
collective is ONE all-gather of the per-rank feature blocks (BASELINE config 4).
Nothing here depends on a GPU being present: the compute step is whatever
callable the caller passes (MfccPlan.clips / .frames in production).
"""
from __future__ import annotations

from typing import Callable, Tuple


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition: rank r owns [r*ceil(N/W), min(N, (r+1)*ceil(N/W)))."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    per = -(-n_items // world) if n_items > 0 else 0
    lo = min(n_items, rank * per)
    hi = min(n_items, lo + per)
    return lo, hi


def gather_features(local, n_items: int, group=None):
    """All-gather the per-rank feature blocks into [n_items, ...] on every rank.

    `local` holds this rank's rows (shard_range order).  all_gather needs equal
    counts, so the short last shard is padded to ceil(N/W) rows and the padding is
    trimmed after the collective.  One collective per call, issued on the current
    stream, so it can overlap the next batch's kernels when the caller runs those
    on another stream.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local
    per = -(-n_items // world)
    if local.shape[0] > per:
        raise ValueError("local block larger than a shard")
    if local.shape[0] < per:
        pad = torch.zeros((per - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], 0)
    out = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out[:n_items]


def sharded_map(compute: Callable, items, group=None, gather: bool = True):
    """items: [N, ...] tensor visible on every rank (or a callable (lo, hi) -> shard that
    produces / loads only this rank's rows).  Runs `compute` on this rank's block and
    returns the gathered [N, ...] features (or just the local block if gather=False)."""
    import torch.distributed as dist

    if dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    if callable(items):
        n = items.n_items
        lo, hi = shard_range(n, rank, world)
        shard = items(lo, hi)
    else:
        n = items.shape[0]
        lo, hi = shard_range(n, rank, world)
        shard = items[lo:hi]
    local = compute(shard)
    if not gather:
        return local
    return gather_features(local, n, group)
