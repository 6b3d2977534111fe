"""Multi-GPU partitioning and the candidate merge.

The search shards without any data-path collective:
  * a batch of DM trials: whole .fft files are dealt round-robin to ranks
    (the reference runs one accelsearch process per file,
    bin/GBNCC_search.py:360-413);
  * one long series: contiguous ranges of the reference's search blocks
    (startr = rlo + k*3735, src/accelsearch.c:165-218), cut so that the ranks have
    equal work; a rank regenerates the plane rows its sub-harmonic look-ups fall
    in instead of receiving them.
The only exchange is an all-gather of fixed-size candidate records (RCCL over
xGMI when the tensors live on GPUs, gloo on CPU for tests).
"""
import numpy as np

REC_FIELDS = ("trial", "r", "z", "sigma", "power", "numharm")


def shard_trials(ntrials, rank, world):
    """Indices of the DM trials rank `rank` searches."""
    return list(range(rank, ntrials, world))


def split_blocks(nblocks, world):
    """Contiguous [first, last) ranges of search blocks, one per rank, sizes differing by <= 1."""
    base, extra = divmod(nblocks, world)
    out, first = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((first, first + n))
        first += n
    return out


# What a (block, z row) costs a rank: generating it (one inverse transform) against summing and thresholding its
# cells, as measured on one MI355X for a 2^27-point, zmax 300, 16-harmonic search (70.4 ms of plane generation over
# 17 968 x 301 rows, 24.2 ms of harmonic sum over 17 965 x 301).
GEN_COST, SUM_COST = 70.4 / (17968 * 301), 24.2 / (17965 * 301)


def range_cost(setup, first, n, need=None):
    """Predicted milliseconds of a rank that searches the search blocks [first, first + n): the rows it generates
    (its own blocks whole; of the blocks its sub-harmonic look-ups fall in, the rows those reach) and the rows it sums."""
    if need is None:
        from .api import block_range_need as need
    if n <= 0:
        return 0.0
    _, rows = need(setup, first, n)
    return GEN_COST * rows + SUM_COST * n * setup.numz


def split_blocks_by_cost(setup, world, need=None):
    """Contiguous [first, last) ranges of search blocks, one per rank, so that every rank has about the same
    WORK -- not the same number of searched blocks: a rank high up the spectrum regenerates most of the plane
    below it (every harmonic h/n of its columns, at rows |z| <= h/n zmax), a rank at the bottom next to nothing.
    The smallest bound T such that cutting greedily from the bottom, each range as long as it costs at most T,
    reaches the top with `world` ranges."""
    nb = setup.num_search_blocks
    cost = lambda a, n: range_cost(setup, a, n, need)

    def cut(T):
        out, first = [], 0
        for _ in range(world):
            lo, hi = 0, nb - first                     # the longest n with cost(first, n) <= T (cost grows with n)
            while lo < hi:
                mid = (lo + hi + 1) // 2
                if cost(first, mid) <= T:
                    lo = mid
                else:
                    hi = mid - 1
            out.append((first, first + lo))
            first += lo
        return out, first

    lo_t, hi_t = 0.0, cost(0, nb)
    for _ in range(40):
        mid = 0.5 * (lo_t + hi_t)
        if cut(mid)[1] >= nb:
            hi_t = mid
        else:
            lo_t = mid
        if hi_t - lo_t <= 1e-3 * hi_t:
            break
    ranges, reached = cut(hi_t)
    assert reached >= nb
    return ranges


def cands_to_records(trial, cands):
    """accelgpu candidate array -> float64 [n, 6] records (trial, r, z, sigma, power, numharm)."""
    rec = np.empty((len(cands), len(REC_FIELDS)), dtype=np.float64)
    rec[:, 0] = trial
    rec[:, 1] = cands["r"]
    rec[:, 2] = cands["z"]
    rec[:, 3] = cands["sigma"]
    rec[:, 4] = cands["power"]
    rec[:, 5] = cands["numharm"]
    return rec


def allgather_rows(rows, device=None):
    """All-gather of variable-length [n, k] float64 arrays (same k on every rank) over
    the default torch.distributed group: counts first, then zero-padded fixed-size
    buffers.  Returns the concatenation in rank order (identical on every rank)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    dev = device if device is not None else torch.device("cpu")
    rows = np.ascontiguousarray(rows, dtype=np.float64)
    ncol = rows.shape[1]
    cnt = torch.tensor([rows.shape[0]], dtype=torch.int64, device=dev)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt)
    counts = [int(c.item()) for c in cnts]
    buf = torch.zeros((max(max(counts), 1), ncol), dtype=torch.float64, device=dev)
    if rows.shape[0]:
        buf[:rows.shape[0]] = torch.from_numpy(rows).to(dev)
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)
    return np.concatenate([b[:n].cpu().numpy() for b, n in zip(bufs, counts)], axis=0)


def allgather_records(records, device=None):
    """The candidate records ([n, 6], see REC_FIELDS) of every rank, in rank order."""
    records = np.asarray(records, dtype=np.float64).reshape(-1, len(REC_FIELDS))
    return allgather_rows(records, device)


def merge_sorted(records):
    """Cross-trial merged list, strongest first (ties by trial then r), as the
    downstream sifting step expects to receive it."""
    if len(records) == 0:
        return records
    order = np.lexsort((records[:, 1], records[:, 0], -records[:, 3]))
    return records[order]


HIT_FIELDS = ("power", "stage", "zind", "block", "rind")


def hits_to_records(hits):
    """accelgpu hit array -> float64 [n, 5]: exact for float32 powers, small ints and
    half-bin columns below 2^53."""
    rec = np.empty((len(hits), len(HIT_FIELDS)), dtype=np.float64)
    for i, f in enumerate(HIT_FIELDS):
        rec[:, i] = hits[f]
    return rec


def records_to_hits(rec, dtype):
    hits = np.zeros(len(rec), dtype=dtype)
    hits["power"] = rec[:, 0].astype(np.float32)
    for i, f in enumerate(HIT_FIELDS[1:], start=1):
        hits[f] = rec[:, i].astype(hits.dtype[f])
    return hits


def reference_order(hits):
    """Sort hits into the order the reference visits cells: block, stage, z row, r."""
    order = np.lexsort((hits["rind"], hits["zind"], hits["stage"], hits["block"]))
    return hits[order]


def search_long_series(gpu, rank, world, max_hits=1 << 22, gather=None):
    """One long series over `world` GPUs: rank searches its contiguous range of
    search blocks (plane blocks it needs are regenerated locally), then the raw
    hits are all-gathered and put back into the reference's order so that the
    candidate list can be replayed exactly as a single process would.
    `gather(records) -> records` defaults to the torch.distributed all-gather."""
    first, last = split_blocks_by_cost(gpu.setup, world)[rank]
    gpu.set_block_range(first, last - first)
    hits = gpu.run(max_hits)
    rec = hits_to_records(hits)
    if gather is None:
        import torch
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
        allrec = allgather_rows(rec, dev)
    else:
        allrec = gather(rec)
    return reference_order(records_to_hits(allrec, hits.dtype))
