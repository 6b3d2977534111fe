#!/usr/bin/env python3
"""Plane generation + harmonic sum in chunks of search blocks (tuning knob hs_chunk_blocks)
against the whole plane at once: ms per search, summed kernel times, same hits.
Usage: python scripts/chunk_sweep.py [chunk sizes ...]"""
import sys
import time

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import presto_amd as pa
from presto_amd import synth

chunks = [int(x) for x in sys.argv[1:]] or [0, 20, 40, 80, 160, 320, 0]
cfg = dict(synth.CONFIGS["cfg2"])
cfg["N"] = 1 << 23
numbins = cfg["N"] // 2
dev = torch.device("cuda", 0)
spectra = []
for i in range(2):
    fft = synth.make_fft(cfg["N"], cfg["dt"], cfg["pulsars"], seed=1000 + i)
    spectra.append(torch.from_numpy(fft.view(np.float32).copy()).to(dev))
torch.cuda.synchronize()
g = pa.AccelSearchGPU(0)
setup = g.configure(numbins, cfg["N"], cfg["dt"], cfg["zmax"], cfg["numharm"], fhi=1e9)
g.set_profiling(True)
ref = None
for cb in chunks:
    g.set_tuning("hs_chunk_blocks", cb)

    def launch(i):
        t = spectra[i % 2]
        g.load_fft_device(t.data_ptr(), numbins, keepalive=t)
        g.run_launch()

    def steps(k):
        launch(0)
        launch(1)
        out = None
        for i in range(k):
            hits = g.run_collect()
            if i + 2 < k:
                launch(i + 2)
            if i == 0:
                out = hits
        return out
    steps(3)
    g.timings(reset=True)
    g.synchronize()
    t0 = time.perf_counter()
    K = 16
    hits = steps(K)
    g.synchronize()
    dt = time.perf_counter() - t0
    tm = g.timings(reset=True)
    if ref is None:
        ref = hits
    same = len(hits) == len(ref) and np.array_equal(hits, ref)
    print("chunk_blocks %4d: %.3f ms/search  correlate %.3f ms  harmsum %.3f ms  launches %d  same hits: %s"
          % (cb, 1e3 * dt / K, tm["correlate"]["total_ms"] / K, tm["harmsum"]["total_ms"] / K,
             tm["harmsum"]["launches"] // K, same), flush=True)
g.close()
