"""CPU, world_size 2, gloo: the N>1 path's sharding + single all-gather.  The
compute step on CPU is the oracle (tests may use it as a stand-in; the product's
kernels need a GPU), so this checks partitioning, padding and gather order."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dsp_amd import dist as D


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 9, 100, 100_000):
        for w in (1, 2, 3, 8):
            spans = [D.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b and c <= d
            per = -(-n // w) if n else 0
            assert all(b - a <= per for a, b in spans)
    # BASELINE config 4: 100 000 clips over 8 GPUs -> 12 500 each
    assert D.shard_range(100_000, 3, 8) == (37_500, 50_000)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_clips, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from tests import signals as S
        clips = torch.from_numpy(np.stack([S.uniform_pm1(1200, 50 + i) for i in range(n_clips)]))

        def compute(shard):
            rows = [O.compute_mfcc(c.numpy(), 500) for c in shard]
            return torch.from_numpy(np.stack(rows)) if rows else torch.zeros((0, 6, 13))

        got = D.sharded_map(compute, clips)
        lo, hi = D.shard_range(n_clips, rank, world)
        q.put((rank, lo, hi, got.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_clips", [5, 4, 1])
def test_two_rank_gather_matches_single_process(n_clips):
    from oracle import oracle as O
    from tests import signals as S
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_clips, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.stack([O.compute_mfcc(S.uniform_pm1(1200, 50 + i), 500) for i in range(n_clips)])
    for rank, lo, hi, got in res:
        assert got.shape == want.shape            # every rank holds the full, trimmed result
        assert np.array_equal(got, want)
    spans = sorted((lo, hi) for _, lo, hi, _ in res)
    assert spans[0][0] == 0 and spans[-1][1] == n_clips


def _pipe_worker(rank, world, port, n_batches, rows, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from tests import signals as S

        def batch(k, r):                                     # rank r's clips of batch k
            return [S.uniform_pm1(1200, 1000 * k + 10 * r + i) for i in range(rows)]

        def compute_into(k):
            def f(out):
                out.copy_(torch.from_numpy(np.stack([O.compute_mfcc(c, 500) for c in batch(k, rank)])))
            return f

        # serial path: compute, gather, next batch
        serial = []
        for k in range(n_batches):
            local = torch.empty((rows, 6, 13))
            compute_into(k)(local)
            serial.append(D.gather_features(local, world * rows).clone())
        # pipelined path: the gather of batch k is in flight while batch k + 1 is computed; results are read one batch late
        pipe = D.GatherPipeline(rows, (6, 13), torch.float32, "cpu")
        piped, pending = [], None
        for k in range(n_batches):
            slot = pipe.submit(compute_into(k))
            if pending is not None:
                piped.append(pipe.result(pending).clone())
            pending = slot
        piped.append(pipe.result(pending).clone())
        pipe.drain()
        q.put((rank, [t.numpy() for t in serial], [t.numpy() for t in piped]))
    finally:
        dist.destroy_process_group()


def test_pipelined_gather_equals_serial_gather_in_order():
    """double-buffered batches, all-gather of batch k overlapping the compute of batch k + 1.  On two gloo ranks
    the pipelined results must be the serial ones, batch by batch (slot reuse after two batches included: 5 batches, depth 2),
    and every rank must hold rank 0's rows first, then rank 1's."""
    from oracle import oracle as O
    from tests import signals as S
    world, port, n_batches, rows = 2, _free_port(), 5, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pipe_worker, args=(r, world, port, n_batches, rows, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, serial, piped in res:
        assert len(serial) == len(piped) == n_batches
        for k in range(n_batches):
            want = np.stack([O.compute_mfcc(S.uniform_pm1(1200, 1000 * k + 10 * r + i), 500) for r in range(world) for i in range(rows)])
            assert np.array_equal(serial[k], want) and np.array_equal(piped[k], want), (rank, k)


def test_pipeline_without_a_process_group_is_the_local_block():
    pipe = D.GatherPipeline(4, (13,), torch.float32, "cpu")
    slot = pipe.submit(lambda out: out.fill_(2.5))
    assert pipe.world == 1 and torch.equal(pipe.result(slot), torch.full((4, 13), 2.5))
    pipe.drain()


def _label_worker(rank, world, port, n_batches, rows, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # BASELINE config 5 at N > 1 (bench.py --workload config5): per clip an int32 label and a float32 probability,
        # packed into one [rows][2] int32 block (the probability's bits) so that ONE all-gather per batch carries both
        def results(k, r):
            lab = torch.tensor([(7 * k + 3 * r + i) % 2 for i in range(rows)], dtype=torch.int32)
            p1 = torch.tensor([0.25 + 0.001 * (100 * k + 10 * r + i) for i in range(rows)], dtype=torch.float32)
            return lab, p1

        pipe = D.GatherPipeline(rows, (2,), torch.int32, "cpu")
        got, pending = [], None
        for k in range(n_batches):
            def compute(block, k=k):
                lab, p1 = results(k, rank)
                block[:, 0].copy_(lab)
                block[:, 1].copy_(p1.view(torch.int32))
            slot = pipe.submit(compute)
            if pending is not None:
                got.append(pipe.result(pending).clone())
            pending = slot
        got.append(pipe.result(pending).clone())
        pipe.drain()
        ok = True
        for k, g in enumerate(got):
            assert g.shape == (world * rows, 2)
            for r in range(world):
                lab, p1 = results(k, r)
                blk = g[r * rows:(r + 1) * rows]
                ok &= bool(torch.equal(blk[:, 0], lab)) and bool(torch.equal(blk[:, 1].contiguous().view(torch.float32), p1))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_config5_label_gather_two_ranks():
    """The per-clip labels + probabilities of the fused clip -> label path gathered on every rank, one packed collective per
    batch, pipelined (5 batches, depth 2): every rank ends with rank 0's clips first, then rank 1's, bit for bit."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_label_worker, args=(r, world, port, 5, 7, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r for r, _ in res) == [0, 1] and all(ok for _, ok in res)
