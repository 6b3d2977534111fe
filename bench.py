#!/usr/bin/env python3
"""Headline benchmark: DM-trial .fft searches per second.

One step = one complete acceleration search of one synthetic DM-trial spectrum
(2^23-point series, zmax 200, numharm 16, sigma 2): block preparation, F-Fdot
plane generation, in-memory harmonic summing, thresholding, hit compaction,
download, ordering and the host-side candidate-list replay.  Spectra are
resident in HBM before the timed region starts.  With N GPUs each rank searches
its own trials (weak scaling, no data-path collective); one RCCL all-gather at
the end of the timed region merges the per-trial candidate lists.

Usage: python bench.py [--gpus N] [--steps K] [--warmup W]
       (N > 1: launched by torch.distributed.run, one rank per GPU)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SUMF2 = {1: 0.0, 2: 0.25, 4: 0.875, 8: 2.1875, 16: 4.84375, 32: 10.171875}
HBM_PEAK_GBS = 8000.0
FP32_VECTOR_PEAK_TFLOPS = 157.3


def cpu_baseline(cfg, threads):
    """Time the CPU oracle ("port" of the reference algorithm) on ONE whole trial
    of the workload (a few seconds on 16 cores)."""
    from oracle import binding as ob
    from presto_amd import synth
    Ns = cfg["N"]
    fft, c = synth.config_fft("cfg2", N=Ns)
    s = ob.Search(fft, Ns, c["dt"], cfg["zmax"], cfg["numharm"], fhi=1e9, nthreads=threads)
    t0 = time.perf_counter()
    hits, cands = s.run()
    dt = time.perf_counter() - t0
    nb_s = s.num_blocks + s.num_prepop
    s.close()
    return dt, nb_s, Ns, len(hits)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (a timed region of K pipelined steps costs K steps plus one fill and drain of the pipeline, 3.6 ms -- the first
    # trial's kernels with nothing beside them, the last trial's download and replay with the GPU idle: at 20 steps
    # that is 4 % of the region, at 100 under 1 %; the first steps after an idle spell run at lower clocks)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--trials", type=int, default=4, help="distinct spectra resident per GPU")
    ap.add_argument("--log2n", type=int, default=23)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N > 1 code path (RCCL process group, candidate all-gather, max over ranks) "
                         "whatever the world size: a one-GPU rehearsal of the multi-GPU launch")
    ap.add_argument("--timed-only", action="store_true",
                    help="skip the legs reported beside the metric (PCIe-inclusive, with optimisation, CPU baseline): "
                         "what a profiler then sees is the timed region alone")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import presto_amd as pa
    from presto_amd import synth
    from presto_amd.api import replay_candidates
    from presto_amd import batch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    multi = world > 1 or args.force_dist
    if multi:
        if "MASTER_ADDR" not in os.environ:            # plain `python bench.py --force-dist`
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=dev)

    cfg = dict(synth.CONFIGS["cfg2"])
    cfg["N"] = 1 << args.log2n
    numbins = cfg["N"] // 2

    # synthetic DM trials, resident in HBM before timing starts
    spectra = []
    for i in range(args.trials):
        fft = synth.make_fft(cfg["N"], cfg["dt"], cfg["pulsars"], seed=1000 + rank * args.trials + i)
        spectra.append(torch.from_numpy(fft.view(np.float32).copy()).to(dev))
    torch.cuda.synchronize()

    g = pa.AccelSearchGPU(local)
    setup = g.configure(numbins, cfg["N"], cfg["dt"], cfg["zmax"], cfg["numharm"], fhi=1e9)
    g.set_profiling(True)

    def launch(i):
        t = spectra[i % len(spectra)]
        g.load_fft_device(t.data_ptr(), numbins, keepalive=t)
        g.run_launch()

    def finish():
        hits = g.run_collect()
        cands, sig, added = replay_candidates(setup, hits)
        return hits, cands

    def run_steps(k):
        """k pipelined steps: download, ordering and host replay of trial i overlap the
        GPU work of trials i+1 and i+2 (two launches in flight, two hit buffers)."""
        out = []
        launch(0)
        if k > 1:
            launch(1)                      # two trials in flight: the GPU never waits for the host
        for i in range(k):
            hits = g.run_collect()
            if i + 2 < k:
                launch(i + 2)
            cands, sig, added = replay_candidates(setup, hits)
            out.append((len(hits), cands))
        return out

    def barrier():
        g.synchronize()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def merge(results):
        # the per-trial candidate lists of all ranks, merged (RCCL all-gather over xGMI)
        recs = [batch.cands_to_records(rank * len(results) + i, c) for i, (_, c) in enumerate(results)]
        return batch.merge_sorted(batch.allgather_records(np.concatenate(recs), dev))

    if args.warmup > 0:
        warm = run_steps(args.warmup)
        if multi:
            merge(warm)             # the collectives' first-use set-up belongs to the warm-up too
    g.timings(reset=True)
    # (the cycle collector stays off inside the timed region, as in timeit: the loop creates no cycles, and a collection
    # of the candidate lists kept for the merge pauses the host for longer than the two trials in flight last --
    # 200 steps: 229-231 searches/s with it, 236-240 without)
    import gc
    gc.disable()
    barrier()
    t0 = time.perf_counter()
    cpu0 = time.process_time()                 # all threads of this process
    results = run_steps(args.steps)
    if multi:
        merged = merge(results)
    barrier()
    elapsed = time.perf_counter() - t0
    host_cpu_s = time.process_time() - cpu0
    # (it stays off through the legs reported beside, which are timed loops of the same kind: a collection of what the
    # timed region left behind, 30 ms, fell into the 40 trials with optimisation on one box in two)
    listed = g.get_tuning("pf_last_listed")    # of the last collected trial: (row, four columns) the coarse bound listed
    bounded = g.get_tuning("hs_bound_active") == 1
    prefilter = bounded and listed != -2       # the coarse bound ran in front of the harmonic sum
    if multi:
        te = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    tm = g.timings(reset=True)

    # PCIe-inclusive rate (reported beside, never as `value`): every trial's spectrum comes
    # from page-locked host memory (where a batch host reads its .fft files into) through
    # accelgpu_load_fft_async(): the copy of trial i+2 runs on the copy engine beside the
    # searches of trials i and i+1, and the launch waits for it on the device.  The same
    # loop with ordinary (pageable) host buffers and the blocking upload is timed as well.
    pcie_rate = pcie_pageable_rate = None
    if rank == 0 and not args.timed_only:
        from presto_amd.api import PinnedSpectrum
        pinned = []
        for sp in spectra[:3]:
            ps = PinnedSpectrum(numbins)
            ps.array[:] = sp.cpu().numpy().view(np.complex64)
            pinned.append(ps)
        kp = max(8, min(args.steps, 20))

        def up(i):
            g.load_fft_async(pinned[i % len(pinned)])
            g.run_launch()

        def upload_steps(k):
            up(0)
            up(1)
            for i in range(k):
                hits = g.run_collect()
                if i + 2 < k:
                    up(i + 2)
                replay_candidates(setup, hits)
            g.synchronize()

        upload_steps(3)                    # the library's own spectrum buffers are allocated on first use
        t1 = time.perf_counter()
        upload_steps(kp)
        pcie_rate = kp / (time.perf_counter() - t1)
        host = [ps.array.copy() for ps in pinned[:2]]
        for ps in pinned:
            ps.free()
        kq = 8
        g.synchronize()
        t1 = time.perf_counter()
        g.load_fft(host[0])
        g.run_launch()
        for i in range(kq):
            if i + 1 < kq:
                g.load_fft(host[(i + 1) % len(host)])
                g.run_launch()
            hits = g.run_collect()
            replay_candidates(setup, hits)
        pcie_pageable_rate = kq / (time.perf_counter() - t1)

    # The whole accelsearch run per trial (reported beside): search, list replay, sort and
    # harmonic elimination, GPU optimisation of every surviving candidate against its own
    # spectrum, pipelined like the search alone: the optimisation of trial i runs beside
    # the search of trial i+1.
    full_rate = opt_ms = n_opt = None
    if rank == 0 and not args.timed_only:
        from presto_amd.api import sort_and_eliminate
        g.set_profiling(False)
        launch(0)
        hits = g.run_collect()
        cands, _, _ = replay_candidates(setup, hits)
        kept = sort_and_eliminate(cands)
        g.optimize(kept[:2])                               # first-use set-up outside the clock
        t2 = time.perf_counter()
        for _ in range(3):
            g.optimize(kept)
        opt_ms = 1e3 * (time.perf_counter() - t2) / 3.0    # one trial's candidates, GPU otherwise idle
        kf = 40                                            # (fill and drain of the two-deep pipeline: 2 % of it)
        g.synchronize()
        t1 = time.perf_counter()
        launch(0)
        launch(1)
        pending = None
        n_opt = 0
        for i in range(kf):
            hits = g.run_collect()
            if i + 2 < kf:
                launch(i + 2)
            cands, _, _ = replay_candidates(setup, hits)
            kept = sort_and_eliminate(cands)
            if pending is not None:
                g.optimize_collect(pending)
            pending = g.optimize_launch(kept)
            n_opt += len(kept)
        g.optimize_collect(pending)
        full_rate = kf / (time.perf_counter() - t1)
        n_opt /= float(kf)
        g.set_profiling(True)

    # The same spectra with every threshold lowered to 0.85 of its value (reported beside): the coarse bound in
    # front of the harmonic sum would then list 2.7 % of its (row, four columns) instead of 0.1 %, which costs more
    # than it saves; a sample taken before the walk says so, the bounded sum runs over the whole plane instead,
    # and the context leaves the coarse bound out of its next trials.  (-sigma goes down to 1, spectra carry
    # interference: the rate must not fall off a cliff there.)
    dense_rate = dense_hits = None
    if rank == 0 and not args.timed_only:
        from presto_amd.api import plan
        dense = plan(numbins, cfg["N"], cfg["dt"], cfg["zmax"], cfg["numharm"], fhi=1e9)
        for st in range(dense.numharmstages):
            dense.powcut[st] *= 0.85
        g.set_search(dense)
        saved_setup, setup = setup, dense
        run_steps(3)
        kd = max(8, min(args.steps, 20))
        g.synchronize()
        t1 = time.perf_counter()
        dres = run_steps(kd)
        g.synchronize()
        dense_rate = kd / (time.perf_counter() - t1)
        dense_hits = float(np.mean([r[0] for r in dres]))
        setup = saved_setup
        g.set_search(setup)
    gc.enable()

    if rank == 0:
        cells = setup.numz * 7470 * setup.num_search_blocks
        nblk = setup.num_search_blocks + setup.num_prepop_blocks
        # The harmonic sum as the reference defines it touches one value per (cell, source): the
        # fundamental and the sub-harmonics' Σ f^2 distinct cells per fundamental cell, 4 B each on a
        # float plane.  The kernels here read the plane's 16-bit upper halves (2 B per value), and with
        # the coarse bound in front (one 16-bit bound per four columns) a quarter of those;
        # both halves only where a bound crosses a threshold.
        float_bytes = 4.0 * (1.0 + SUMF2[cfg["numharm"]]) * cells
        hs_alg = float_bytes / 8 if prefilter else float_bytes / 2 if bounded else float_bytes
        hs = tm["harmsum"]
        hs_ms = hs["total_ms"] / max(hs["launches"], 1)
        co = tm["correlate"]
        co_ms = co["total_ms"] / max(co["launches"], 1)
        # plane generation: the plane has to be stored (4 B per cell of every block) and the spectrum and the
        # template bank have to be read once; the group maxima it writes beside the plane for the coarse bound
        # (2 B per four cells) are this design's own overhead, not work the reference's formulation asks for
        L = setup.fftlen
        co_alg = 4.0 * setup.numz * 7470 * nblk + 8.0 * numbins + 8.0 * setup.numz * L
        co_overhead = 2.0 * setup.numz * 1868 * nblk if prefilter else 0.0
        co_flop = nblk * setup.numz * (5.0 * L * np.log2(L) + 6.0 * L + 3.0 * 7470)
        pmc = {}
        pmc_path = os.path.join(ROOT, "profiles", "pmc_kernels.json")
        if os.path.exists(pmc_path) and args.log2n == 23:
            # PMC counters cannot be read inside a timed run: these are the committed rocprofv3
            # measurements of the same kernels on the same workload
            pmc = json.load(open(pmc_path))
            if pmc.get("prefilter", False) != prefilter or pmc.get("bounded", False) != bounded:
                pmc = {}
        src = "profiles/pmc_kernels.json (rocprofv3 --pmc passes of this command, not this run)" if pmc else None
        hs_name = ("k_prefilter_lanes + k_flagged_exact <%d stages> over one 16-bit bound per four columns (windows of eight)" if prefilter
                   else "k_harmsum_bw<%d stages> over 16-bit upper halves of the plane" if bounded
                   else "k_harmsum<%d stages>") % setup.numharmstages
        roof_hs = {"kernel": hs_name, "bound": "hbm", "achieved": hs_alg / (hs_ms * 1e-3) / 1e9 if hs_ms > 0 else 0.0,
                   "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "traffic": pmc.get("harmsum_hbm_bytes_per_launch"), "traffic_source": src,
                   "algorithmic_bytes_per_launch": hs_alg, "avg_launch_ms": hs_ms,
                   # the same launch priced as a read of the float plane (4 B per value: what the reference's
                   # formulation moves, and what the kernels of the earlier rounds were priced with)
                   "float_plane_bytes_per_launch": float_bytes,
                   "float_plane_equivalent_GBps": float_bytes / (hs_ms * 1e-3) / 1e9 if hs_ms > 0 else 0.0}
        roof_hs["frac"] = roof_hs["achieved"] / HBM_PEAK_GBS
        if prefilter:
            # what the coarse bound's time depends on: the share of its (row, four columns) that it had to list for
            # exact sums (-1: its list was of no use and the bounded sum ran over the whole plane)
            groups = float(setup.numz) * setup.num_search_blocks * 1868
            roof_hs["listed"] = listed
            roof_hs["listed_fraction"] = listed / groups if listed >= 0 else None
        # Plane generation is FFT arithmetic on the vector ALUs (no matrix product in it, so no MFMA): what binds it is
        # fp32 vector instruction issue, and it is priced against the 157.3 TFLOP/s fp32 vector peak.  Its compulsory
        # bytes against the HBM peak are reported beside (they come to the same fraction on this workload).
        co_tflops = co_flop / (co_ms * 1e-3) / 1e12 if co_ms > 0 else 0.0
        roof_co = {"kernel": "k_correlate_rr (plane generation: %d inverse transforms of %d points)" % (nblk * setup.numz, L),
                   "bound": "fp32-vector", "achieved": co_tflops, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                   "frac": co_tflops / FP32_VECTOR_PEAK_TFLOPS,
                   "flop_per_launch": co_flop, "avg_launch_ms": co_ms,
                   "traffic": pmc.get("correlate_hbm_bytes_per_launch"), "traffic_source": src,
                   "algorithmic_bytes_per_launch": co_alg, "overhead_bytes_per_launch": co_overhead,
                   "hbm_GBps_on_algorithmic_bytes": co_alg / (co_ms * 1e-3) / 1e9 if co_ms > 0 else 0.0,
                   "frac_of_hbm_peak": co_alg / (co_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if co_ms > 0 else 0.0}
        # `roofline` describes the kernel that takes most of the step; the other one is reported beside it
        dominant, other = (roof_co, roof_hs) if co_ms >= hs_ms else (roof_hs, roof_co)
        line = {
            "metric": "DM-trial .fft searches/sec (2^%d pts, zmax=%d, numharm=%d)"
                      % (args.log2n, cfg["zmax"], cfg["numharm"]),
            "value": args.steps * world / elapsed,
            "unit": "searches/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: accelsearch -zmax 200 -numharm 16 -sigma 2 on one "
                                   "synthetic 2^%d-point .fft per step, in-HBM ffdot plane" % args.log2n,
                       "numbins": numbins, "numz": setup.numz, "fftlen": setup.fftlen,
                       "search_blocks": setup.num_search_blocks, "plane_GB": setup.plane_bytes / 1e9,
                       "trials_per_gpu": args.steps, "parallelism": "dm-trials x%d" % world},
            "roofline": dominant,
            "roofline_other_kernel": other,
            "kernels_ms_per_step": {k: v["total_ms"] / max(args.steps, 1) for k, v in tm.items()},
            "hits_per_trial": float(np.mean([r[0] for r in results])),
            "cands_per_trial": float(np.mean([len(r[1]) for r in results])),
            "pcie_inclusive_searches_per_s": pcie_rate,
            "pcie_inclusive_pageable_searches_per_s": pcie_pageable_rate,
            "dense_spectrum_searches_per_s": dense_rate,
            "dense_spectrum_hits_per_trial": dense_hits,
            "with_optimization_searches_per_s": full_rate,
            "optimize_ms_per_trial": opt_ms,
            "cands_optimized_per_trial": n_opt,
            # what a rank asks of its host: CPU seconds of all its threads per trial of the timed region (launches,
            # ordering the hits, sigma per hit on up to 8 threads when a trial has thousands, the list replay)
            "host_cpu_s_per_trial": host_cpu_s / max(args.steps, 1),
            "host_threads": {"main": 1, "sigma_workers_max": 7},
        }
        if world == 1 and not args.no_cpu and not args.timed_only:
            threads = min(16, os.cpu_count() or 1)
            dt_s, nb_s, Ns, nh = cpu_baseline(cfg, threads)
            nb_full = setup.num_search_blocks + setup.num_prepop_blocks
            line["cpu_baseline"] = {
                "value": 1.0 / (dt_s * nb_full / nb_s), "unit": "searches/s", "cores": threads,
                "kind": "port",
                "sample": "CPU oracle (C port of the reference algorithm, own FFT, OpenMP over z) on "
                          "one whole 2^%d-point trial with the same flags: %d of %d blocks in %.2f s"
                          % (int(np.log2(Ns)), nb_s, nb_full, dt_s)}
        print(json.dumps(line))
    g.close()
    if multi:
        dist.barrier()          # rank 0 reports a little longer; leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
