"""Plane generation with outputs stored singly or in adjacent pairs, with and without the 16-bit
bounds: kernel times, same hits, same plane bits.  python scripts/corr_pair_ab.py [cfg2|cfg4]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import presto_amd as pa                      # noqa: E402
from presto_amd import synth                 # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
numbins, dt, zmax, nh = {"cfg2": (1 << 22, 64e-6, 200, 16), "cfg4": (1 << 26, 64e-6, 300, 16)}[name]
g = pa.AccelSearchGPU(0)
fft = synth.make_fft(2 * numbins, dt, dict(synth.CONFIGS["cfg2"])["pulsars"], seed=1000) if name == "cfg2" else synth.noise_fft(numbins, 5)
setup = g.configure(numbins, 2 * numbins, dt, zmax, nh, fhi=1e9)
g.load_fft(fft)
g.set_profiling(True)
ref = None
for rep in range(2):
    for pair, bound in ((0, 0), (1, 0), (0, 1), (1, 1)):
        g.set_tuning("corr_pair", pair)
        g.set_tuning("hs_bound", bound)
        hits = g.run(max_hits=1 << 22)
        g.timings()
        for _ in range(3):
            hits = g.run(max_hits=1 << 22)
        tm = g.timings()
        rows = np.concatenate([g.get_plane(z, 2 * setup.rlo, 7470 * 40) for z in (0, setup.numz // 2, setup.numz - 1)])
        if ref is None:
            ref = (hits, rows)
        same = hits.tobytes() == ref[0].tobytes() and rows.tobytes() == ref[1].tobytes()
        print("%s pairs=%d bounds=%d: correlate %.3f ms harmsum %.3f ms, %d hits, same hits and plane bits as the first: %s" % (
            name, pair, bound, tm["correlate"]["total_ms"] / 3, tm["harmsum"]["total_ms"] / 3, len(hits), same), flush=True)
g.close()
