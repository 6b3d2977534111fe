"""Harmonic sum over the 16-bit bound plane against the plain one: same hits, kernel times.
python scripts/hs_bound_ab.py [cfg2 ...]   (one GPU)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import presto_amd as pa                      # noqa: E402
from presto_amd import synth                 # noqa: E402

SUMF2 = {1: 0.0, 2: 0.25, 4: 0.875, 8: 2.1875, 16: 4.84375, 32: 10.171875}
CASES = {"cfg1": (1 << 19, 1e-4, 0, 8), "cfg2": (1 << 22, 64e-6, 200, 16), "cfg2h8": (1 << 22, 64e-6, 200, 8),
         "cfg2h4": (1 << 22, 64e-6, 200, 4), "cfg2h2": (1 << 22, 64e-6, 200, 2), "cfg2h1": (1 << 22, 64e-6, 200, 1), "cfg2s": (1 << 20, 64e-6, 50, 16),
         "cfg4": (1 << 26, 64e-6, 300, 16)}
g = pa.AccelSearchGPU(0)
for name in sys.argv[1:] or ["cfg2s", "cfg2"]:
    numbins, dt, zmax, nh = CASES[name]
    if name in ("cfg2", "cfg2s"):
        cfg = dict(synth.CONFIGS["cfg2"])
        fft = synth.make_fft(2 * numbins, dt, cfg["pulsars"], seed=1000)
    else:
        fft = synth.noise_fft(numbins, 5)
    setup = g.configure(numbins, 2 * numbins, dt, zmax, nh, fhi=1e9)
    g.load_fft(fft)
    g.set_profiling(True)
    res = {}
    for mode in (0, 1, 2, 3):
        g.set_tuning("hs_bound", mode)
        hits = g.run(max_hits=1 << 22)
        g.timings()
        for _ in range(3):
            hits = g.run(max_hits=1 << 22)
        tm = g.timings()
        hs = tm["harmsum"]["total_ms"] / 3
        alg = 4.0 * (1 + SUMF2[nh]) * setup.numz * 7470 * setup.num_search_blocks
        print("%s hs_bound=%d: correlate %.3f ms harmsum %.3f ms (%.0f GB/s of float-plane bytes), %d hits" % (
            name, mode, tm["correlate"]["total_ms"] / 3, hs, alg / hs / 1e6, len(hits)), flush=True)
        res.setdefault(mode, hits)
    same = all(len(res[0]) == len(res[m]) and res[0].tobytes() == res[m].tobytes() for m in res)
    print("%s: hits identical: %s" % (name, same), flush=True)
g.close()
