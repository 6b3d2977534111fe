#!/usr/bin/env python3
"""How much does running plane generation (VALU/LDS bound) beside the harmonic sum (HBM
bound) buy?  Upper-bound experiment: K contexts on one GPU, each with its own plane and
streams, each running the pipelined search loop of bench.py from its own host thread.
Prints searches/s for K = 1, 2, 3."""
import sys
import threading
import time

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import presto_amd as pa
from presto_amd import synth
from presto_amd.api import replay_candidates

cfg = dict(synth.CONFIGS["cfg2"])
cfg["N"] = 1 << 23
numbins = cfg["N"] // 2
dev = torch.device("cuda", 0)
spectra = []
for i in range(4):
    fft = synth.make_fft(cfg["N"], cfg["dt"], cfg["pulsars"], seed=1000 + i)
    spectra.append(torch.from_numpy(fft.view(np.float32).copy()).to(dev))
torch.cuda.synchronize()


def worker(g, setup, k, out, idx):
    def launch(i):
        t = spectra[i % len(spectra)]
        g.load_fft_device(t.data_ptr(), numbins, keepalive=t)
        g.run_launch()
    launch(0)
    launch(1)
    for i in range(k):
        hits = g.run_collect()
        if i + 2 < k:
            launch(i + 2)
        replay_candidates(setup, hits)
    g.synchronize()
    out[idx] = k


for K in (1, 2, 3):
    ctxs = []
    for _ in range(K):
        g = pa.AccelSearchGPU(0)
        setup = g.configure(numbins, cfg["N"], cfg["dt"], cfg["zmax"], cfg["numharm"], fhi=1e9)
        ctxs.append((g, setup))
    for steps in (4, 24):          # warm-up, then timed
        out = [0] * K
        th = [threading.Thread(target=worker, args=(g, s, steps, out, i)) for i, (g, s) in enumerate(ctxs)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
    print("contexts %d: %.1f searches/s (%.2f ms per search)" % (K, sum(out) / dt, 1e3 * dt / sum(out)), flush=True)
    for g, _ in ctxs:
        g.close()
