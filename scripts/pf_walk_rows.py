"""Walk time against rows walked (timing only): python scripts/pf_walk_rows.py"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import presto_amd as pa                      # noqa: E402
from presto_amd import synth                 # noqa: E402

nh = 16
numbins, dt, zmax = 1 << 22, 64e-6, 200
g = pa.AccelSearchGPU(0)
fft = synth.make_fft(2 * numbins, dt, dict(synth.CONFIGS["cfg2"])["pulsars"], seed=1000)
setup = g.configure(numbins, 2 * numbins, dt, zmax, nh, fhi=1e9)
g.load_fft(fft)
g.set_profiling(True)
g.set_tuning("hs_bound", 4)
g.set_tuning("pf_sample", 0)
g.set_tuning("pf_groups", 1)
for nt in (256, 512, 1024):
    g.set_tuning("pf_nt", nt)
    for rows in (50, 201):
        g.set_tuning("pf_rows", rows)
        g.run(max_hits=1 << 23)
        g.timings()
        for _ in range(3):
            g.run(max_hits=1 << 23)
        tm = g.timings()
        print("nt %d rows %3d: harmsum %.3f ms, listed %d" % (nt, rows, tm["harmsum"]["total_ms"] / 3, g.get_tuning("pf_listed")), flush=True)
g.close()
