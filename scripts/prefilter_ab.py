"""The coarse bound (group maxima; hs_bound=1 uses it where it pays) in front of the bounded harmonic sum
against the bounded sum alone (hs_bound=3): same hits, kernel times, how many (row, group)s were listed.
python scripts/prefilter_ab.py [cfg2 ...]   (one GPU)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import presto_amd as pa                      # noqa: E402
from presto_amd import synth                 # noqa: E402

CASES = {"cfg2": (1 << 22, 64e-6, 200, 16), "cfg2h8": (1 << 22, 64e-6, 200, 8), "cfg2h32": (1 << 22, 64e-6, 200, 32),
         "cfg2s": (1 << 20, 64e-6, 50, 16), "cfg2noise": (1 << 22, 64e-6, 200, 16)}
g = pa.AccelSearchGPU(0)
for name in sys.argv[1:] or ["cfg2s", "cfg2"]:
    numbins, dt, zmax, nh = CASES[name]
    if name in ("cfg2", "cfg2s"):
        fft = synth.make_fft(2 * numbins, dt, dict(synth.CONFIGS["cfg2"])["pulsars"], seed=1000)
    else:
        fft = synth.noise_fft(numbins, 5)
    setup = g.configure(numbins, 2 * numbins, dt, zmax, nh, fhi=1e9)
    g.load_fft(fft)
    g.set_profiling(True)
    res = {}
    for mode in (3, 1, 3, 1):
        g.set_tuning("hs_bound", mode)
        hits = g.run(max_hits=1 << 22)
        g.timings()
        for _ in range(3):
            hits = g.run(max_hits=1 << 22)
        tm = g.timings()
        listed = g.get_tuning("pf_listed") if mode == 1 else 0
        groups = setup.numz * setup.num_search_blocks * 1868
        print("%s hs_bound=%d: correlate %.3f ms harmsum %.3f ms, %d hits, listed %d of %d (row, group)s = %.3f %%" % (
            name, mode, tm["correlate"]["total_ms"] / 3, tm["harmsum"]["total_ms"] / 3, len(hits), listed, groups,
            100.0 * listed / groups), flush=True)
        res.setdefault(mode, hits)
    print("%s: hits identical: %s" % (name, res[1].tobytes() == res[3].tobytes()), flush=True)
g.close()
