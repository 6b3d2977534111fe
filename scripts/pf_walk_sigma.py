"""Time of the coarse bound (sample off, list forced) against the search's sigma, i.e. against the length of the list:
python scripts/pf_walk_sigma.py"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import presto_amd as pa                      # noqa: E402
from presto_amd import synth                 # noqa: E402

nh = 16
numbins, dt, zmax = 1 << 22, 64e-6, 200
g = pa.AccelSearchGPU(0)
fft = synth.make_fft(2 * numbins, dt, dict(synth.CONFIGS["cfg2"])["pulsars"], seed=1000)
g.load_fft(fft)
g.set_profiling(True)
for sigma in (12.0, 5.0, 3.0, 2.0):
    setup = g.configure(numbins, 2 * numbins, dt, zmax, nh, sigma=sigma, fhi=1e9)
    g.set_tuning("hs_bound", 4)
    g.set_tuning("pf_sample", 0)
    g.fill_plane()
    g.search_plane(max_hits=1 << 23)
    g.timings()
    for _ in range(5):
        hits = g.search_plane(max_hits=1 << 23)
    tm = g.timings()
    print("sigma %4.1f: harmsum %.3f ms, %d hits, listed %d" % (sigma, tm["harmsum"]["total_ms"] / 5, len(hits), g.get_tuning("pf_listed")), flush=True)
g.close()
