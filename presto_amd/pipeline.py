"""Batch driver: many DM-trial files over the GPUs of one node, sifted together.

What a survey pipeline does per beam around accelsearch (reference
bin/GBNCC_search.py:360-413: one `zapbirds`, `rednoise`, `accelsearch` process
per DM trial, then `sifting` over all ACCEL files), as one program per GPU:

    python -m torch.distributed.run --nproc-per-node N -m presto_amd.pipeline \
        -zmax 200 -numharm 16 [-zaplist birds.zaplist] [-rednoise] -o beam  *.fft

Rank r takes files r, r+N, ... ; for each it loads the spectrum (or transforms the
.dat time series on the device), optionally zaps and de-reddens it in HBM, searches,
trims and optimises the candidates on the GPU and writes <root>_ACCEL_<zmax>,
.cand and .txtcand exactly as the stand-alone program does.  The files move through
a three-stage pipeline: a reader thread holds the next file in memory, the GPU
searches the current one, and the host finishes the previous one (list replay,
device optimisation beside the running search, result files).  The fundamentals of
all files are then all-gathered (RCCL over xGMI; gloo on CPU) and rank 0 runs the
cross-DM sifting and writes <out>.accelcands and <out>.report.
"""
import argparse
import os
import sys
import time

import numpy as np

from . import batch, sifting

# seconds spent per step of the driver since the process started (diagnostics)
TIMES = {}


class _timed:
    def __init__(self, key):
        self.key = key

    def __enter__(self):
        self.t0 = time.perf_counter()

    def __exit__(self, *exc):
        TIMES[self.key] = TIMES.get(self.key, 0.0) + time.perf_counter() - self.t0


# one fixed-size record per fundamental for the exchange
SIFT_FIELDS = ("file", "candnum", "sigma", "numharm", "ipow", "cpow", "r", "z", "snr", "dm")
MAX_HARM = 32


def read_inf(root):
    """N and dt from a PRESTO .inf file."""
    N = dt = None
    with open(root + ".inf") as f:
        for line in f:
            if line.startswith(" Number of bins in the time series"):
                N = int(float(line.split("=")[-1]))
            elif line.startswith(" Width of each time series bin (sec)"):
                dt = float(line.split("=")[-1])
    if N is None or dt is None:
        raise ValueError("cannot read N and dt from %s.inf" % root)
    return N, dt


class _Trial:
    """One input file on its way through the GPU."""
    __slots__ = ("index", "path", "root", "ext", "N", "dt", "data", "numbins", "setup", "t0", "nopt", "kept", "pinned", "work_file")


class PinnedPool:
    """Page-locked host buffers for the spectra on their way to the GPU: the reader thread reads a file
    straight into one, the upload from it is asynchronous (the copy engine works beside the running
    searches and the host does not wait), and the buffer comes back when its trial has been collected.
    One set of `count` buffers per spectrum length, made when the first file of that length turns up."""

    def __init__(self, count=6):
        import queue
        self._queue = queue
        self.count = count
        self.free = {}

    def get(self, numbins):
        from . import api
        if numbins not in self.free:
            q = self._queue.Queue()
            for _ in range(self.count):
                q.put(api.PinnedSpectrum(numbins))
            self.free[numbins] = q
        return self.free[numbins].get()

    def put(self, ps):
        self.free[ps.array.size].put(ps)

    def close(self):
        for q in self.free.values():
            while not q.empty():
                q.get().free()
        self.free = {}


def read_trial(index, path, pool=None):
    """Host side of the input: the .inf numbers and the samples (no GPU call; runs on the
    reader thread, the read releases the interpreter lock).  With a `pool`, a spectrum is read
    into page-locked memory."""
    t = _Trial()
    t.index, t.path, t.t0 = index, path, time.perf_counter()
    t.root, t.ext = os.path.splitext(path)
    t.N, t.dt = read_inf(t.root)
    t.pinned = t.work_file = None
    if t.ext == ".fft" and pool is not None:
        t.numbins = os.path.getsize(path) // 8
        t.pinned = pool.get(t.numbins)
        t.data = t.pinned.array
        with open(path, "rb", buffering=0) as f:
            got = f.readinto(memoryview(t.data).cast("B"))
        if got != 8 * t.numbins:
            raise ValueError("short read on %s" % path)
    elif t.ext == ".fft":
        t.data = np.fromfile(path, dtype=np.complex64)
        t.numbins = t.data.size
    elif t.ext in (".dat", ".sdat"):
        t.data = np.fromfile(path, dtype=np.int16 if t.ext == ".sdat" else np.float32).astype(np.float32)
        t.numbins = t.data.size // 2
    else:
        raise ValueError("input file ('%s') must be an '.fft' or '.[s]dat' file" % path)
    return t


def _geometry(t, opts):
    return (t.numbins, t.N, t.dt, opts.zmax, opts.numharm, opts.sigma, opts.flo, opts.fhi, bool(opts.locpow))


def stage_trial(gpu, t, opts, state):
    """Upload, prepare and launch one trial.  Everything here runs beside a search that
    is still in flight (the library keeps several spectrum buffers); only a change of
    the search geometry needs the GPU drained first, which the caller does."""
    geo = _geometry(t, opts)
    if state.get("geometry") != geo:
        state["setup"] = gpu.configure(t.numbins, t.N, t.dt, opts.zmax, opts.numharm, sigma=opts.sigma,
                                       flo=opts.flo, fhi=opts.fhi, norm_type=1 if opts.locpow else 0)
        state["geometry"] = geo
    t.setup = state["setup"]
    T = t.N * t.dt
    dat_input = t.ext != ".fft"
    with _timed("upload+prepare"):
        if dat_input:
            gpu.load_dat(t.data, deredden=True)
            gpu.set_bin0(1.0, 1.0)          # no -photon here (src/accel_utils.c:1553-1556)
        elif t.pinned is not None:
            gpu.load_fft_async(t.pinned)    # no host wait: the launch waits for the copy on the device
        else:
            gpu.load_fft(t.data)
        t.data = None
        if opts.birds is not None:
            gpu.zapbirds(*opts.birds(T))
        if opts.rednoise and not dat_input:
            gpu.rednoise(T)
    with _timed("launch"):
        gpu.run_launch()


def collect_trial(gpu, t, opts, writer=None, pool=None):
    """First part of what follows a launch: collect the hits, replay the candidate list,
    write the work file (on the `writer` executor if there is one) and trim the list."""
    from . import api
    setup = t.setup
    with _timed("collect"):
        hits = gpu.run_collect(max_hits=1 << 18)
    if t.pinned is not None and pool is not None:
        pool.put(t.pinned)                  # the search has read its spectrum on the device: the host buffer is free
        t.pinned = None
    with _timed("replay"):
        cands, sig, added = api.replay_candidates(setup, hits)
    accelnm = "%s_ACCEL_%d" % (t.root, opts.zmax)    # as given: named before zmax is rounded up (src/accel_utils.c:1492-1494, 1526)
    if t.ext == ".fft":                     # the work file of the search (src/accel_utils.c:1291-1294)
        with _timed("txtcand"):
            if writer is None:
                api.write_txtcand(setup, hits, sig, added, accelnm + ".txtcand")
            else:
                t.work_file = writer.submit(api.write_txtcand, setup, hits, sig, added, accelnm + ".txtcand")
    t.nopt = 0
    t.kept = None
    if len(cands):
        with _timed("trim"):
            t.kept = api.sort_and_eliminate(cands, eliminate=opts.numharm > 1 and not opts.noharmremove)


def optimise_trial(gpu, t, opts):
    """Second part: queue the optimisation of the trimmed list on the device (against this trial's
    spectrum, beside the next trial's search).  One optimisation is out at a time: the one before
    has been collected (write_trial)."""
    if t.kept is not None:
        with _timed("optimize launch"):
            t.nopt = gpu.optimize_launch(t.kept, polish=not opts.noharmpolish)
        t.kept = None


def write_trial(gpu, t, opts, writer=None):
    """Third part: wait for the optimisation, write <root>_ACCEL_<zmax> and .cand, and
    return the sifting Candlist of the trial's fundamentals (before any rejection); with
    a `writer` executor the files are written there and a future is returned."""
    from . import api
    setup = t.setup
    accelnm = "%s_ACCEL_%d" % (t.root, opts.zmax)    # as given: named before zmax is rounded up (src/accel_utils.c:1492-1494, 1526)
    opt = None
    if t.nopt:
        with _timed("optimize wait"):
            opt = gpu.optimize_collect(t.nopt)

    def write_results():
        if t.work_file is not None:
            t.work_file.result()            # (queued before this on the same worker: done; a failure surfaces here)
        if opt is None:
            return sifting.Candlist([])
        with _timed("result files"):
            with open(t.root + ".inf") as f:
                inf_text = f.read()
            api.write_accel_files(setup, opt, accelnm, accelnm + ".cand", inf_text)
            m = sifting.DM_re.search(accelnm)
            return sifting.candlist_from_optcands(opt, setup, m.groups()[0] if m else "0.00", accelnm)
    return write_results() if writer is None else writer.submit(write_results)


def search_one_file(gpu, path, opts):
    """The whole accelsearch run on one file, start to end.  Returns the Candlist of its
    fundamentals and the seconds spent."""
    t = read_trial(0, path)
    stage_trial(gpu, t, opts, {})
    collect_trial(gpu, t, opts)
    optimise_trial(gpu, t, opts)
    cl = write_trial(gpu, t, opts)
    return cl, time.perf_counter() - t.t0


def search_files(gpu, files, indices, opts, depth=2):
    """Generator over (index, Candlist, seconds since the previous result) for the files
    `indices`, as a three-stage pipeline: a reader thread has the next file in memory
    while the GPU searches the current one and the host finishes the previous one."""
    import queue
    import threading
    from concurrent.futures import ThreadPoolExecutor
    q = queue.Queue(maxsize=depth)
    pool = PinnedPool(depth + 4)            # queued + being read + two in flight + one spare

    def reader():
        try:
            gpu.bind_thread()               # page-locked memory is asked of the context's device
            for i in indices:
                q.put(read_trial(i, files[i], pool))
            q.put(None)
        except BaseException as e:          # hand the failure to the consumer
            q.put(e)

    th = threading.Thread(target=reader, daemon=True)
    th.start()
    writer = ThreadPoolExecutor(max_workers=1)      # result files, off the launch path
    state = {}
    launched = None                                 # trial whose search is queued or running
    optimising = None                               # trial whose candidates are being optimised
    last = time.perf_counter()
    written = []                                    # (index, future of the Candlist), in order

    def ready(all_of_them=False):
        nonlocal last
        while written and (all_of_them or written[0][1].done()):
            i, fut = written.pop(0)
            cl = fut.result()
            now = time.perf_counter()
            yield i, cl, now - last
            last = now

    def advance():
        """One step for the trials behind the newest launch.  The older search is collected first: the
        optimisation queued a step ago ran beside it on the device and is over by the time its results
        are asked for (waiting for it before the search put its whole length, 3 ms of dependent steps,
        on the host's path: 192 files/s instead of 2xx)."""
        nonlocal launched, optimising
        if launched is not None:
            collect_trial(gpu, launched, opts, writer, pool)
        if optimising is not None:
            written.append((optimising.index, write_trial(gpu, optimising, opts, writer)))
            optimising = None
        if launched is not None:
            optimise_trial(gpu, launched, opts)
            optimising, launched = launched, None

    # The cycle collector stays off while files are in flight: the loop creates no reference cycles, and a full
    # collection of the candidate lists kept for sifting pauses the host for longer than the two searches in
    # flight last (a 200-trial loop on one GPU: 229-231 trials/s with it, 236-240 without).
    import gc
    gc_was_on = gc.isenabled()
    gc.disable()
    try:
        while True:
            with _timed("wait for reader"):
                t = q.get()
            if isinstance(t, BaseException):
                raise t
            if t is None:
                break
            if launched is not None and state.get("geometry") != _geometry(t, opts):
                advance()                           # a new geometry rebuilds the plane: drain first
            prev, launched = launched, None
            stage_trial(gpu, t, opts, state)        # queued behind prev's search on the GPU
            launched = prev
            advance()                               # host work and optimisation overlap t's search
            launched = t
            yield from ready()
        advance()
        advance()
        yield from ready(all_of_them=True)
    finally:
        writer.shutdown(wait=True)
        if gc_was_on:
            gc.enable()
    th.join()
    gpu.synchronize()
    pool.close()


def candlist_to_records(cl, file_index):
    """Candlist -> float64 [n, 10 + MAX_HARM] (SIFT_FIELDS, then the harmonic powers)."""
    rec = np.zeros((len(cl), len(SIFT_FIELDS) + MAX_HARM))
    for i, c in enumerate(cl):
        rec[i, :10] = (file_index, c.candnum, c.sigma, c.numharm, c.ipow_det, c.cpow, c.r, c.z, c.snr, c.DM)
        rec[i, 10:10 + c.numharm] = c.harm_pows
    return rec


def records_to_candlist(rec, filenames, T, track=True):
    """The inverse, on the rank that sifts."""
    cands = []
    for row in rec:
        fn = filenames[int(row[0])]
        c = sifting.Candidate(int(row[1]), float(row[2]), int(row[3]), float(row[4]), float(row[5]),
                              float(row[6]), float(row[7]), "%.2f" % row[9], fn, T[int(row[0])])
        c.harm_pows = np.array(row[10:10 + c.numharm])
        c.snr = float(row[8])
        c.hits = [(c.DM, c.snr, c.sigma)]
        cands.append(c)
    return sifting.Candlist(cands, trackbad=track, trackdupes=track)


def sift(candlist, dmstrs, min_dm_hits=2, low_dm_cutoff=2.0, verbosity=1):
    """The sequence of bin/ACCEL_sift.py: default rejections, duplicates, DM problems,
    harmonics, sorted by sigma."""
    candlist.default_rejection()
    if len(candlist):
        candlist.remove_duplicate_candidates(verbosity)
    if len(candlist):
        candlist.remove_DM_problems(min_dm_hits, dmstrs, low_dm_cutoff, verbosity)
    if len(candlist) > 1:
        candlist.remove_harmonics(verbosity)
    candlist.cands.sort(key=lambda c: (-c.sigma, -c.ipow_det))
    return candlist


def make_bird_fn(zaplist, baryv):
    """T -> (lobins, hibins) of the zap list, read once per observation length."""
    if not zaplist:
        return None
    from . import api
    cache = {}

    def birds(T):
        if T not in cache:
            cache[T] = api.read_birds(zaplist, T, baryv)
        return cache[T]
    return birds


def run(files, opts, rank=0, world=1, device=0, gather=None):
    """Search this rank's share of `files`, gather, sift on rank 0.  Returns
    (sifted Candlist or None, per-file seconds of this rank)."""
    from . import api
    mine = batch.shard_trials(len(files), rank, world)
    gpu = api.AccelSearchGPU(device)
    recs, secs, Ts = [], [], {}
    for i, cl, dt_s in search_files(gpu, files, mine, opts):
        recs.append(candlist_to_records(cl, i))
        secs.append(dt_s)
    gpu.close()
    mine_rec = np.concatenate(recs) if recs else np.zeros((0, len(SIFT_FIELDS) + MAX_HARM))
    allrec = mine_rec if world == 1 else (gather or batch.allgather_rows)(mine_rec, None)
    if rank != 0:
        return None, secs
    for i, f in enumerate(files):
        N, dt = read_inf(os.path.splitext(f)[0])
        Ts[i] = N * dt
    names = ["%s_ACCEL_%d" % (os.path.splitext(f)[0], opts.zmax) for f in files]
    order = np.lexsort((allrec[:, 1], allrec[:, 0]))          # by file, then candidate number
    cl = records_to_candlist(allrec[order], names, Ts)
    dmstrs = []
    for n in names:
        m = sifting.DM_re.search(n)
        dmstrs.append(m.groups()[0] if m else "0.00")
    cl = sift(cl, dmstrs, opts.min_dm_hits, opts.low_dm_cutoff, verbosity=opts.verbosity)
    if opts.out:
        cl.to_file(opts.out + ".accelcands")
        cl.write_cand_report(opts.out + ".report")
    return cl, secs


def parse_args(argv):
    ap = argparse.ArgumentParser(prog="presto_amd.pipeline", prefix_chars="-", allow_abbrev=False)
    ap.add_argument("-zmax", type=int, default=200)
    ap.add_argument("-numharm", type=int, default=8)
    ap.add_argument("-sigma", type=float, default=2.0)
    ap.add_argument("-flo", type=float, default=1.0)
    ap.add_argument("-fhi", type=float, default=10000.0)
    ap.add_argument("-locpow", action="store_true")
    ap.add_argument("-noharmpolish", action="store_true")
    ap.add_argument("-noharmremove", action="store_true")
    ap.add_argument("-zaplist", default=None)
    ap.add_argument("-baryv", type=float, default=0.0)
    ap.add_argument("-rednoise", action="store_true")
    ap.add_argument("-min_dm_hits", type=int, default=2)
    ap.add_argument("-low_dm_cutoff", type=float, default=2.0)
    ap.add_argument("-verbosity", type=int, default=1)
    ap.add_argument("-o", dest="out", default=None)
    ap.add_argument("files", nargs="+")
    opts = ap.parse_args(argv)
    opts.birds = make_bird_fn(opts.zaplist, opts.baryv)
    return opts


def main(argv=None):
    opts = parse_args(sys.argv[1:] if argv is None else argv)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    gather = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        dist.init_process_group("nccl", device_id=dev)
        gather = lambda rec, _d: batch.allgather_rows(rec, dev)   # noqa: E731
    t0 = time.perf_counter()
    cl, secs = run(sorted(opts.files), opts, rank, world, local, gather)
    if rank == 0:
        print("%d files on %d GPU(s) in %.2f s; %d candidates after sifting" % (
            len(opts.files), world, time.perf_counter() - t0, len(cl)))
        cl.print_cand_summary()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
