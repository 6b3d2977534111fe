"""Bounded against float-plane harmonic sum as hits get dense (thresholds lowered by hand): cfg2 plane.
python scripts/hs_bound_dense.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import presto_amd as pa                      # noqa: E402
from presto_amd import synth                 # noqa: E402
from presto_amd.api import plan              # noqa: E402

numbins, dt, zmax, nh = 1 << 22, 64e-6, 200, 16
g = pa.AccelSearchGPU(0)
fft = synth.noise_fft(numbins, 5)
g.load_fft(fft)
g.set_tuning("hits_cap", 1 << 26)
for scale in (1.0, 0.85, 0.75, 0.65, 0.55):
    setup = plan(numbins, 2 * numbins, dt, zmax, nh, fhi=1e9)
    for s in range(setup.numharmstages):
        setup.powcut[s] *= scale
    g.create_subharminfos(setup.zmax)
    g.set_search(setup)
    g.set_profiling(True)
    out = []
    for mode in (0, 1):
        g.set_tuning("hs_bound", mode)
        hits = g.run(max_hits=1 << 26)
        g.timings()
        hits = g.run(max_hits=1 << 26)
        tm = g.timings()
        out.append("hs_bound=%d %.2f ms" % (mode, tm["harmsum"]["total_ms"]))
    print("thresholds x %.2f: %9d hits; harmonic sum %s" % (scale, len(hits), ", ".join(out)), flush=True)
g.close()
