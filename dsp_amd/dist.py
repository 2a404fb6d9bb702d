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


class GatherPipeline:
    """Double-buffered all-gather of per-batch feature blocks (BASELINE config 4): the collective of batch k
    runs on the backend's own stream (RCCL's, over xGMI) while the kernels of batch k + 1 run on the caller's stream.

        pipe = GatherPipeline(rows_per_rank, row_shape, dtype, device)
        for k in range(K):
            slot = pipe.submit(lambda out: plan.clips(batch[k], 500, out))    # compute into the slot's local block, gather async
            ...                                                              # next iteration's compute overlaps this gather
            full = pipe.result(slot)        # when the consumer needs batch k: [world * rows_per_rank, ...] on every rank
        pipe.drain()

    With depth = 2 a slot's buffers are reused two batches later; submit() first waits for the gather that last used the
    slot (on the GPU that is a stream dependency, not a host block: torch's Work.wait() for NCCL makes the current stream
    wait for the collective).  Results are those of the serial path -- the same collective on the same bytes, only the
    order of issue relative to the next batch's kernels changes.  Every rank contributes exactly rows_per_rank rows
    (all_gather needs equal counts: pad the short last shard, see gather_features)."""

    def __init__(self, rows_per_rank: int, row_shape, dtype, device, group=None, depth: int = 2, always_collective: bool = False):
        """always_collective: issue the all-gather even in a one-rank group (a copy through the backend): lets a one-GPU box
        exercise the RCCL path itself -- communicator, the backend's stream, Work.wait() -- which world == 1 otherwise skips."""
        import torch
        import torch.distributed as dist
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.collective = self.world > 1 or (always_collective and dist.is_initialized())
        self.depth = depth
        shape = (rows_per_rank,) + tuple(row_shape)
        self.local = [torch.empty(shape, dtype=dtype, device=device) for _ in range(depth)]
        # world == 1: nothing to gather, the local block IS the result
        self.gathered = [torch.empty((self.world * rows_per_rank,) + tuple(row_shape), dtype=dtype, device=device) if self.collective
                         else self.local[i] for i in range(depth)]
        self.work = [None] * depth
        self.k = 0

    def submit(self, compute: Callable) -> int:
        """compute(local_block) fills this rank's rows of the next batch (enqueued on the current stream); the all-gather of
        that batch is issued asynchronously right behind it.  Returns the slot to pass to result()."""
        import torch.distributed as dist
        slot = self.k % self.depth
        self._wait(slot)                                   # the gather that last read local[slot] / wrote gathered[slot]
        compute(self.local[slot])
        if self.collective:
            self.work[slot] = dist.all_gather_into_tensor(self.gathered[slot], self.local[slot], group=self.group, async_op=True)
        self.k += 1
        return slot

    def _wait(self, slot: int):
        if self.work[slot] is not None:
            self.work[slot].wait()
            self.work[slot] = None

    def result(self, slot: int):
        """The gathered [world * rows_per_rank, ...] block of the batch submitted into `slot` (valid until the slot is reused)."""
        self._wait(slot)
        return self.gathered[slot]

    def drain(self):
        for s in range(self.depth):
            self._wait(s)


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
