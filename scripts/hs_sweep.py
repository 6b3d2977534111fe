"""Times the harmonic-sum kernel alone on a cfg2-sized plane (env knobs select variants)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import presto_amd as pa
from presto_amd import synth
numbins = 1 << 22
nh = int(sys.argv[1]) if len(sys.argv) > 1 else 16
zmax = int(sys.argv[2]) if len(sys.argv) > 2 else 200
fft = synth.noise_fft(numbins, 3)
g = pa.AccelSearchGPU(0)
setup = g.configure(numbins, 2 * numbins, 64e-6, zmax, nh, fhi=1e9)
g.load_fft(fft)
g.fill_plane()
g.set_profiling(True)
g.search_plane(); g.timings()
for _ in range(5):
    h = g.search_plane()
t = g.timings()["harmsum"]
ms = t["total_ms"] / t["launches"]
S = {1: 0, 2: .25, 4: .875, 8: 2.1875, 16: 4.84375, 32: 10.171875}[nh]
alg = 4 * (1 + S) * setup.numz * 7470 * setup.num_search_blocks
print("J=%s remap=%s nh=%d zmax=%d: %.3f ms  alg %.2f GB -> %.0f GB/s (%.1f%% of 8 TB/s) hits %d" % (
    os.environ.get("ACCELGPU_HS_J", "0"), "off" if os.environ.get("ACCELGPU_HS_XCD_REMAP") == "0" else "on",
    nh, zmax, ms, alg / 1e9, alg / ms / 1e6, alg / ms / 1e6 / 80, len(h)))
