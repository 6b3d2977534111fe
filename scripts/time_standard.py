"""Times the standard (not in-memory) search mode on the benchmark configuration."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import presto_amd as pa
from presto_amd import synth
N = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 23)
fft, c = synth.config_fft("cfg2", N=N)
g = pa.AccelSearchGPU(0)
g.configure(fft.size, c["N"], c["dt"], c["zmax"], c["numharm"], fhi=1e9, inmem=False)
g.load_fft(fft)
hits = g.run_standard()
t = time.perf_counter()
for _ in range(3):
    hits = g.run_standard()
dt = (time.perf_counter() - t) / 3
print("standard mode, N=2^%d, zmax %d, numharm %d: %.1f ms per search (%d hits)" % (
    int(np.log2(N)), c["zmax"], c["numharm"], dt * 1e3, len(hits)))
g.configure(fft.size, c["N"], c["dt"], c["zmax"], c["numharm"], fhi=1e9, inmem=True)
g.load_fft(fft)
hits = g.run()
t = time.perf_counter()
for _ in range(3):
    hits = g.run()
print("in-memory mode: %.1f ms per search (%d hits)" % ((time.perf_counter() - t) / 3 * 1e3, len(hits)))
