"""The large configurations with and without the coarse bound in front of the bounded harmonic sum:
hs_bound = 3 (bounded sum alone), 1 (the library's choice), 4 (coarse bound forced).
python scripts/prefilter_big.py [cfg4 cfg5 ...]   (one GPU, up to 190 GB of HBM)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import presto_amd as pa                      # noqa: E402
from presto_amd import synth                 # noqa: E402

CASES = {"cfg2": (1 << 22, 64e-6, 200, 16), "cfg5": (1 << 23, 64e-6, 1200, 32), "cfg4": (1 << 26, 64e-6, 300, 16),
         "cfg2h8": (1 << 22, 64e-6, 200, 8), "cfg2h32": (1 << 22, 64e-6, 200, 32),
         # 16384-point transforms (zmax above 300) with up to 16 harmonics
         "z400h16": (1 << 22, 64e-6, 400, 16), "z400h8": (1 << 22, 64e-6, 400, 8), "z400h4": (1 << 22, 64e-6, 400, 4), "z1200h16": (1 << 22, 64e-6, 1200, 16),
         "cfg5h16": (1 << 23, 64e-6, 1200, 16)}
g = pa.AccelSearchGPU(0)
for name in sys.argv[1:] or ["cfg4", "cfg5"]:
    numbins, dt, zmax, nh = CASES[name]
    fft = synth.noise_fft(numbins, 5)
    res = {}
    for mode in (3, 1, 4):
        g.set_tuning("hs_bound", mode)
        setup = g.configure(numbins, 2 * numbins, dt, zmax, nh, fhi=1e9)
        g.load_fft(fft)
        g.set_profiling(True)
        hits = g.run(max_hits=1 << 22)
        g.timings()
        hits = g.run(max_hits=1 << 22)
        tm = g.timings()
        print("%s hs_bound=%d (coarse bound %s): plane %.1f GB, correlate %.2f ms, harmsum %.2f ms, %d hits, listed %d" % (
            name, mode, "on" if g.get_tuning("hs_prefilter_active") else "off", setup.plane_bytes / 1e9,
            tm["correlate"]["total_ms"], tm["harmsum"]["total_ms"], len(hits), g.get_tuning("pf_listed")), flush=True)
        res[mode] = hits
    print("%s: hits identical: %s" % (name, all(res[3].tobytes() == h.tobytes() for h in res.values())), flush=True)
g.close()
