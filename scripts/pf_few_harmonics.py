"""Few harmonics on the cfg2 plane: the bounded sum (hs_bound = 3) against the coarse bound forced (hs_bound = 4,
two adjacent groups per lane): harmonic-sum time per search and hits.  python scripts/pf_few_harmonics.py (one GPU)"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import presto_amd as pa                      # noqa: E402
from presto_amd import synth                 # noqa: E402
g = pa.AccelSearchGPU(0)
N, dt = 1 << 23, 64e-6
fft = synth.make_fft(N, dt, dict(synth.CONFIGS["cfg2"])["pulsars"], seed=1000)
for numharm in (2, 4, 8):
    g.configure(N // 2, N, dt, 200, numharm, fhi=1e9)
    g.load_fft(fft)
    g.set_profiling(True)
    ref = None
    for mode in (3, 4, 3, 4):
        g.set_tuning("hs_bound", mode)
        h = g.run(max_hits=1 << 22)
        g.timings()
        for _ in range(3):
            h = g.run(max_hits=1 << 22)
        tm = g.timings()
        ref = h if ref is None else ref
        print("numharm %d hs_bound=%d: correlate %.3f ms harmsum %.3f ms, %d hits, same %s" % (
            numharm, mode, tm["correlate"]["total_ms"] / 3, tm["harmsum"]["total_ms"] / 3, len(h), h.tobytes() == ref.tobytes()), flush=True)
g.close()
