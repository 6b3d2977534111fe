"""Bounded harmonic sum against workgroups per CU (dynamic LDS caps them): python scripts/hs_bound_occ.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import presto_amd as pa                      # noqa: E402
from presto_amd import synth                 # noqa: E402

numbins, dt, zmax, nh = 1 << 22, 64e-6, 200, 16
g = pa.AccelSearchGPU(0)
fft = synth.make_fft(2 * numbins, dt, dict(synth.CONFIGS["cfg2"])["pulsars"], seed=1000)
setup = g.configure(numbins, 2 * numbins, dt, zmax, nh, fhi=1e9)
g.load_fft(fft)
g.set_profiling(True)
g.set_tuning("hs_bound", int(sys.argv[1]) if len(sys.argv) > 1 else 2)   # 2: one column per lane, 3: several
for wgs in (8, 7, 6, 5, 4, 3):
    g.set_tuning("hs_lds", 0 if wgs == 8 else (160 * 1024) // wgs - 512)
    g.run(max_hits=1 << 22)
    g.timings()
    for _ in range(3):
        hits = g.run(max_hits=1 << 22)
    tm = g.timings()
    print("at most %d workgroups per CU: harmsum %.3f ms, %d hits" % (wgs, tm["harmsum"]["total_ms"] / 3, len(hits)), flush=True)
g.close()
