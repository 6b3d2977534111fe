"""The bounded sum with one column per lane (hs_bound = 2) against several (3) for few harmonics, on a
2^23-point zmax 200 plane: python scripts/hs_wide_few.py"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import presto_amd as pa                      # noqa: E402
from presto_amd import synth                 # noqa: E402

numbins, dt, zmax = 1 << 22, 64e-6, 200
g = pa.AccelSearchGPU(0)
fft = synth.noise_fft(numbins, 5)
for nh in (1, 2, 4, 8):
    res = {}
    for mode in (2, 3):
        g.set_tuning("hs_bound", mode)
        g.configure(numbins, 2 * numbins, dt, zmax, nh, fhi=1e9)
        g.load_fft(fft)
        g.set_profiling(True)
        g.run(max_hits=1 << 22)
        g.timings()
        for _ in range(3):
            hits = g.run(max_hits=1 << 22)
        tm = g.timings()
        res[mode] = hits
        print("numharm %2d hs_bound=%d: harmsum %.3f ms, %d hits" % (nh, mode, tm["harmsum"]["total_ms"] / 3, len(hits)), flush=True)
    print("numharm %2d: hits identical: %s" % (nh, res[2].tobytes() == res[3].tobytes()), flush=True)
g.close()
