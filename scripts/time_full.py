"""Where the time of one whole accelsearch run goes (search, replay, trimming, optimisation)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch                                  # noqa: E402
import presto_amd as pa                       # noqa: E402
from presto_amd import synth                  # noqa: E402
from presto_amd.api import replay_candidates, sort_and_eliminate   # noqa: E402

cfg = dict(synth.CONFIGS["cfg2"])
numbins = cfg["N"] // 2
dev = torch.device("cuda", 0)
spectra = []
for i in range(3):
    fft = synth.make_fft(cfg["N"], cfg["dt"], cfg["pulsars"], seed=1000 + i)
    spectra.append(torch.from_numpy(fft.view(np.float32).copy()).to(dev))
g = pa.AccelSearchGPU(0)
setup = g.configure(numbins, cfg["N"], cfg["dt"], cfg["zmax"], cfg["numharm"], fhi=1e9)
T = {k: 0.0 for k in ("launch", "collect", "replay", "trim", "optimize")}
n = 0
for it in range(8):
    t = spectra[it % 3]
    t0 = time.perf_counter()
    g.load_fft_device(t.data_ptr(), numbins, keepalive=t)
    g.run_launch()
    t1 = time.perf_counter()
    hits = g.run_collect()
    t2 = time.perf_counter()
    cands, _, _ = replay_candidates(setup, hits)
    t3 = time.perf_counter()
    kept = sort_and_eliminate(cands)
    t4 = time.perf_counter()
    opt = g.optimize(kept)
    t5 = time.perf_counter()
    if it >= 2:
        n += 1
        for k, a, b in (("launch", t0, t1), ("collect", t1, t2), ("replay", t2, t3), ("trim", t3, t4), ("optimize", t4, t5)):
            T[k] += b - a
print("hits %d cands %d kept %d" % (len(hits), len(cands), len(kept)))
print("  ".join("%s %.2f ms" % (k, 1e3 * v / n) for k, v in T.items()), " total %.2f ms" % (1e3 * sum(T.values()) / n))
