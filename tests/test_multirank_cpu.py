"""World-size-2 tests of the multi-rank path on CPU (gloo): trial sharding,
block-range splitting and the all-gather merge of candidate records."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from presto_amd import batch


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ntrials, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng_rows = []
    for t in batch.shard_trials(ntrials, rank, world):
        rng = np.random.default_rng(100 + t)
        n = int(rng.integers(0, 6))
        c = np.zeros(n, dtype=[("power", "f4"), ("sigma", "f4"), ("numharm", "i4"),
                               ("added_order", "i4"), ("r", "f8"), ("z", "f8")])
        c["r"], c["z"] = rng.uniform(10, 1e5, n), rng.integers(-100, 100, n) * 2.0
        c["sigma"], c["power"], c["numharm"] = rng.uniform(2, 30, n), rng.uniform(20, 500, n), 1 << rng.integers(0, 5, n)
        rng_rows.append(batch.cands_to_records(t, c))
    local = np.concatenate(rng_rows) if rng_rows else np.zeros((0, 6))
    merged = batch.merge_sorted(batch.allgather_records(local))
    q.put((rank, merged))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_split():
    assert batch.shard_trials(10, 1, 4) == [1, 5, 9]
    assert sorted(sum((batch.shard_trials(1024, r, 8) for r in range(8)), [])) == list(range(1024))
    r = batch.split_blocks(17965, 8)
    assert r[0][0] == 0 and r[-1][1] == 17965 and all(a[1] == b[0] for a, b in zip(r, r[1:]))
    assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1
    assert batch.split_blocks(3, 8)[3:] == [(3, 3)] * 5


def test_split_by_cost_balances_the_ranks():
    """One long series (2^27 points, zmax 300, 16 harmonics) over 2, 4 and 8 ranks: ranges cut for equal predicted
    work are contiguous, cover every search block, and leave the slowest rank within 15 % of the mean; equal
    searched ranges do not (the top rank regenerates half the plane, the bottom one an eighth)."""
    from presto_amd.api import plan, block_range_need
    N = 1 << 27
    s = plan(N // 2, N, 64e-6, 300, 16, fhi=1e9)
    nb = s.num_search_blocks
    blocks, rows = block_range_need(s, 0, nb)
    # the whole range: every block; the blocks below the search region only serve as sources, and no harmonic reaches
    # their outermost rows
    assert blocks == nb + s.num_prepop_blocks and nb * s.numz < rows <= blocks * s.numz
    assert block_range_need(s, 0, 0) == (0, 0)
    for world in (2, 4, 8):
        r = batch.split_blocks_by_cost(s, world)
        assert r[0][0] == 0 and r[-1][1] == nb and all(a[1] == b[0] for a, b in zip(r, r[1:]))
        cost = [batch.range_cost(s, a, b - a) for a, b in r]
        assert max(cost) <= 1.15 * sum(cost) / world
        assert max(cost) < batch.range_cost(s, 0, nb) / (1.0 + 0.3 * np.log2(world))   # and it is worth splitting
    even = [batch.range_cost(s, a, b - a) for a, b in batch.split_blocks(nb, 8)]
    assert max(even) > 1.3 * sum(even) / 8
    # a source-only block is generated for the rows its harmonics reach, not whole: the top eighth's need
    blocks, rows = block_range_need(s, nb - nb // 8, nb // 8)
    assert rows < 0.7 * blocks * s.numz


@pytest.mark.timeout(120)
def test_allgather_merge_world2():
    world, ntrials = 2, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ntrials, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=100) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # every rank holds the same merged list, and it is what one process would build
    assert np.array_equal(got[0], got[1])
    rows = []
    for t in range(ntrials):
        rng = np.random.default_rng(100 + t)
        n = int(rng.integers(0, 6))
        rows.append(n)
    assert got[0].shape == (sum(rows), 6)
    assert set(got[0][:, 0].astype(int)) == {t for t, n in enumerate(rows) if n}
    assert np.all(np.diff(got[0][:, 3]) <= 0)          # strongest first


HIT_DTYPE = np.dtype([("power", "f4"), ("stage", "i4"), ("zind", "i4"), ("block", "i4"), ("rind", "i8")])


def _fake_hits(nblocks, seed):
    rng = np.random.default_rng(seed)
    n = 500
    h = np.zeros(n, dtype=HIT_DTYPE)
    h["block"] = rng.integers(0, nblocks, n)
    h["stage"] = rng.integers(0, 5, n)
    h["zind"] = rng.integers(0, 201, n)
    h["rind"] = 1072 + h["block"].astype(np.int64) * 7470 + rng.integers(0, 7470, n)
    h["power"] = rng.uniform(20, 3000, n).astype(np.float32)
    return batch.reference_order(h)


def _range_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    allhits = _fake_hits(37, 5)
    from presto_amd.api import plan
    real = plan(37 * 3735 + 3000, 2 * (37 * 3735 + 3000), 1e-3, 20, 8, flo=0.2, fhi=1e9)   # host arithmetic only
    assert real.num_search_blocks == 37
    first, last = batch.split_blocks_by_cost(real, world)[rank]
    mine = allhits[(allhits["block"] >= first) & (allhits["block"] < last)]

    class FakeGPU:                      # stands in for AccelSearchGPU on a CPU-only box
        setup = real

        def set_block_range(self, a, n):
            assert (a, n) == (first, last - first)

        def run(self, max_hits):
            return mine

    merged = batch.search_long_series(FakeGPU(), rank, world)
    q.put((rank, merged))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_long_series_hit_merge_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_range_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=100) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = _fake_hits(37, 5)
    assert np.array_equal(got[0], want) and np.array_equal(got[1], want)


def _sift_worker(rank, world, port, q):
    import torch.distributed as dist
    from presto_amd import batch, pipeline, sifting
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T = 500.0
    # four DM trials dealt round-robin; a pulsar at r = 20000 seen at DM 10, 20, 30
    lists = {}
    for i, dm in enumerate([0.0, 10.0, 20.0, 30.0]):
        cands = []
        if dm > 0:
            c = sifting.Candidate(1, 12.0 + (dm == 20.0), 4, 60.0, 200.0, 20000.0 + 0.1 * i, 2.0,
                                  "%.2f" % dm, "b_DM%.2f_ACCEL_50" % dm, T)
            c.harm_pows = np.array([30.0, 15.0, 10.0, 9.0]); c.harms_to_snr(); c.hits = [(dm, c.snr, c.sigma)]
            cands.append(c)
        c = sifting.Candidate(2, 9.0, 2, 30.0, 150.0, 40000.0 + 0.2, 4.0, "%.2f" % dm,
                              "b_DM%.2f_ACCEL_50" % dm, T)
        c.harm_pows = np.array([20.0, 12.0]); c.harms_to_snr(); c.hits = [(dm, c.snr, c.sigma)]
        cands.append(c)
        lists[i] = sifting.Candlist(cands)
    mine = batch.shard_trials(4, rank, world)
    rec = np.concatenate([pipeline.candlist_to_records(lists[i], i) for i in mine])
    allrec = batch.allgather_rows(rec, None)
    if rank == 0:
        names = ["b_DM%.2f_ACCEL_50" % dm for dm in (0.0, 10.0, 20.0, 30.0)]
        order = np.lexsort((allrec[:, 1], allrec[:, 0]))
        cl = pipeline.records_to_candlist(allrec[order], names, {i: T for i in range(4)})
        cl = pipeline.sift(cl, ["0.00", "10.00", "20.00", "30.00"], 2, 2.0, verbosity=0)
        q.put([(c.DM, round(c.r), len(c.hits), c.sigma) for c in cl])
    dist.destroy_process_group()


def test_sifting_after_gather_world2():
    """Fundamentals found on two ranks are all-gathered and sifted on rank 0: the pulsar
    collects its three DM hits; the candidate that peaks at DM 0 is dropped; the 2nd
    harmonic goes too."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_sift_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert out == [(20.0, 20000, 3, 13.0)]
