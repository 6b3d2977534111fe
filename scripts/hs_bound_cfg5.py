"""32 harmonics (six stages) on the 80 GB plane of cfg5, or a smaller stand-in: the float-plane
register-lean sum against the bounded one.  python scripts/hs_bound_cfg5.py [log2 numbins] [zmax]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import presto_amd as pa                      # noqa: E402
from presto_amd import synth                 # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 23
zmax = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
numbins = 1 << lg
g = pa.AccelSearchGPU(0)
fft = synth.noise_fft(numbins, 5)
res = {}
for mode in (0, 1, 0, 1):
    g.set_tuning("hs_bound", mode)
    setup = g.configure(numbins, 2 * numbins, 64e-6, zmax, 32, fhi=1e9)
    g.load_fft(fft)
    g.set_profiling(True)
    hits = g.run(max_hits=1 << 22)
    g.timings()
    hits = g.run(max_hits=1 << 22)
    tm = g.timings()
    print("2^%d bins zmax %d numharm 32, hs_bound=%d (active %d): plane %.1f GB, correlate %.1f ms, harmsum %.1f ms, %d hits" % (
        lg, zmax, mode, g.get_tuning("hs_bound_active"), setup.plane_bytes / 1e9, tm["correlate"]["total_ms"],
        tm["harmsum"]["total_ms"], len(hits)), flush=True)
    res.setdefault(mode, hits)
print("hits identical:", res[0].tobytes() == res[1].tobytes())
g.close()
