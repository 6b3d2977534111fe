"""Times plane generation (block prep + correlate) alone on a cfg2-sized spectrum."""
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import presto_amd as pa
from presto_amd import synth
numbins = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 22)
zmax = int(sys.argv[1]) if len(sys.argv) > 1 else 200
fft = synth.noise_fft(numbins, 3)
g = pa.AccelSearchGPU(0)
setup = g.configure(numbins, 2 * numbins, 64e-6, zmax, 1, fhi=1e9)
g.load_fft(fft)
g.fill_plane()
g.set_profiling(True)
for _ in range(4):
    g.fill_plane()
t = g.timings()
nfft = setup.numz * (setup.num_search_blocks + setup.num_prepop_blocks)
ms = t["correlate"]["total_ms"] / t["correlate"]["launches"]
L = setup.fftlen
flop = nfft * (5 * L * np.log2(L) + 6 * L + 3 * 7470)
print("zmax=%d fftlen=%d: correlate %.3f ms (%d FFTs, %.1f ns/FFT, %.1f TFLOP/s), prep %.3f ms, plane store %.0f GB/s" % (
    zmax, L, ms, nfft, ms * 1e6 / nfft, flop / ms / 1e9, t["block_prep"]["total_ms"] / t["block_prep"]["launches"],
    setup.numz * 7470 * 4.0 * (setup.num_search_blocks + setup.num_prepop_blocks) / ms / 1e6))
