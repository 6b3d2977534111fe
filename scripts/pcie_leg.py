#!/usr/bin/env python3
"""The pipelined search loop with every spectrum uploaded from page-locked host memory
(accelgpu_load_fft_async) against the same loop on spectra resident in HBM."""
import sys
import time

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import presto_amd as pa
from presto_amd import synth
from presto_amd.api import PinnedSpectrum, replay_candidates

cfg = dict(synth.CONFIGS["cfg2"])
cfg["N"] = 1 << 23
numbins = cfg["N"] // 2
dev = torch.device("cuda", 0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
pinned, resident = [], []
for i in range(3):
    fft = synth.make_fft(cfg["N"], cfg["dt"], cfg["pulsars"], seed=1000 + i)
    ps = PinnedSpectrum(numbins)
    ps.array[:] = fft
    pinned.append(ps)
    resident.append(torch.from_numpy(fft.view(np.float32).copy()).to(dev))
torch.cuda.synchronize()
g = pa.AccelSearchGPU(0)
setup = g.configure(numbins, cfg["N"], cfg["dt"], cfg["zmax"], cfg["numharm"], fhi=1e9)


def loop(up, k):
    up(0)
    up(1)
    for i in range(k):
        hits = g.run_collect()
        if i + 2 < k:
            up(i + 2)
        replay_candidates(setup, hits)
    g.synchronize()


def up_pinned(i):
    g.load_fft_async(pinned[i % 3])
    g.run_launch()


def up_resident(i):
    t = resident[i % 3]
    g.load_fft_device(t.data_ptr(), numbins, keepalive=t)
    g.run_launch()


for name, up in (("resident", up_resident), ("pinned", up_pinned), ("resident", up_resident), ("pinned", up_pinned)):
    loop(up, 3)
    t0 = time.perf_counter()
    loop(up, K)
    dt = time.perf_counter() - t0
    print("%-9s %.3f ms/search = %.1f searches/s" % (name, 1e3 * dt / K, K / dt), flush=True)
g.close()
