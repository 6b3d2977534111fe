"""Batch driver: many DM-trial files over the GPUs of one node, sifted together.

What a survey pipeline does per beam around accelsearch (reference
bin/GBNCC_search.py:360-413: one `zapbirds`, `rednoise`, `accelsearch` process
per DM trial, then `sifting` over all ACCEL files), as one program per GPU:

    python -m torch.distributed.run --nproc-per-node N -m presto_amd.pipeline \
        -zmax 200 -numharm 16 [-zaplist birds.zaplist] [-rednoise] -o beam  *.fft

Rank r takes files r, r+N, ... ; for each it loads the spectrum (or transforms the
.dat time series on the device), optionally zaps and de-reddens it in HBM, searches,
trims and optimises the candidates on the GPU and writes <root>_ACCEL_<zmax>,
.cand and .txtcand exactly as the stand-alone program does.  The fundamentals of
all files are then all-gathered (RCCL over xGMI; gloo on CPU) and rank 0 runs the
cross-DM sifting and writes <out>.accelcands and <out>.report.
"""
import argparse
import os
import sys
import time

import numpy as np

from . import batch, sifting

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


def search_one_file(gpu, path, opts):
    """The whole accelsearch run on one file.  Returns the sifting Candlist of its
    fundamentals (before any rejection) and the seconds spent."""
    from . import api
    t0 = time.perf_counter()
    root, ext = os.path.splitext(path)
    N, dt = read_inf(root)
    T = N * dt
    dat_input = ext in (".dat", ".sdat")
    if ext == ".fft":
        spec = np.fromfile(path, dtype=np.complex64)
        numbins = spec.size
    elif dat_input:
        series = np.fromfile(path, dtype=np.int16 if ext == ".sdat" else np.float32).astype(np.float32)
        numbins = series.size // 2
    else:
        raise ValueError("input file ('%s') must be an '.fft' or '.[s]dat' file" % path)
    setup = gpu.configure(numbins, N, dt, opts.zmax, opts.numharm, sigma=opts.sigma, flo=opts.flo,
                          fhi=opts.fhi, norm_type=1 if opts.locpow else 0)
    if dat_input:
        gpu.load_dat(series, deredden=True)
    else:
        gpu.load_fft(spec)
    if opts.birds is not None:
        gpu.zapbirds(*opts.birds(T))
    if opts.rednoise and not dat_input:
        gpu.rednoise(T)
    hits, cands, sig, added = gpu.search()
    if not dat_input:                       # the work file of the search (src/accel_utils.c:1291-1294)
        with open("%s_ACCEL_%d.txtcand" % (root, setup.zmax), "w") as f:
            for h, s in zip(hits[added != 0], sig[added != 0]):
                nh = 1 << int(h["stage"])
                rr = h["rind"] * 0.5 / nh
                f.write("%-7.2f  %-7.4f  %-2d  %-14.4f  %-14.9f  %-10.4f\n" % (
                    h["power"], s, nh, rr, rr / T, (setup.zlo + h["zind"] * 2.0) / nh))
    accelnm = "%s_ACCEL_%d" % (root, setup.zmax)
    if len(cands) == 0:
        return sifting.Candlist([]), time.perf_counter() - t0
    kept = api.sort_and_eliminate(cands, eliminate=opts.numharm > 1 and not opts.noharmremove)
    opt = gpu.optimize(kept, polish=not opts.noharmpolish)
    api.write_accel_files(setup, opt, accelnm, accelnm + ".cand", open(root + ".inf").read())
    m = sifting.DM_re.search(accelnm)
    cl = sifting.candlist_from_optcands(opt, setup, m.groups()[0] if m else "0.00", accelnm)
    return cl, time.perf_counter() - t0


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
    from . import api
    return (lambda T: api.read_birds(zaplist, T, baryv)) if zaplist else None


def run(files, opts, rank=0, world=1, device=0, gather=None):
    """Search this rank's share of `files`, gather, sift on rank 0.  Returns
    (sifted Candlist or None, per-file seconds of this rank)."""
    from . import api
    mine = batch.shard_trials(len(files), rank, world)
    gpu = api.AccelSearchGPU(device)
    recs, secs, Ts = [], [], {}
    for i in mine:
        cl, dt_s = search_one_file(gpu, files[i], opts)
        recs.append(candlist_to_records(cl, i))
        secs.append(dt_s)
    gpu.close()
    mine_rec = np.concatenate(recs) if recs else np.zeros((0, len(SIFT_FIELDS) + MAX_HARM))
    allrec = mine_rec if world == 1 else (gather or batch._allgather_generic)(mine_rec, None)
    if rank != 0:
        return None, secs
    for i, f in enumerate(files):
        N, dt = read_inf(os.path.splitext(f)[0])
        Ts[i] = N * dt
    names = ["%s_ACCEL_%d" % (os.path.splitext(f)[0], opts.zmax + opts.zmax % 2) for f in files]
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
        gather = lambda rec, _d: batch._allgather_generic(rec, dev)   # noqa: E731
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
