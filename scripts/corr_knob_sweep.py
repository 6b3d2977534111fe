"""Plane generation (with bounds, paired stores) against its launch-shape knobs on cfg2.
python scripts/corr_knob_sweep.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import presto_amd as pa                      # noqa: E402
from presto_amd import synth                 # noqa: E402

numbins, dt, zmax, nh = 1 << 22, 64e-6, 200, 16
g = pa.AccelSearchGPU(0)
fft = synth.noise_fft(numbins, 5)
setup = g.configure(numbins, 2 * numbins, dt, zmax, nh, fhi=1e9)
g.load_fft(fft)
g.set_profiling(True)


def timeit(label):
    g.run(max_hits=1 << 22)
    g.timings()
    for _ in range(4):
        g.run(max_hits=1 << 22)
    tm = g.timings()
    print("%-34s correlate %.3f ms" % (label, tm["correlate"]["total_ms"] / 4), flush=True)


timeit("defaults (chunk 16, tile 4 x 16)")
for chunk in (8, 12, 24, 32):
    g.set_tuning("corr_chunk", chunk)
    timeit("chunk %d" % chunk)
g.set_tuning("corr_chunk", 0)
for tz, tc in ((2, 16), (8, 16), (4, 8), (4, 32), (2, 32), (8, 8), (1, 64)):
    g.set_tuning("corr_tz", tz)
    g.set_tuning("corr_tc", tc)
    timeit("tile %d x %d" % (tz, tc))
g.set_tuning("corr_tz", -1)
g.set_tuning("corr_tc", 16)
timeit("defaults again")
g.close()
