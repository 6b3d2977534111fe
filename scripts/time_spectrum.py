"""Times time-series input (upload + long FFT + unpack + deredden) on the device."""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import presto_amd as pa
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 23)
rng = np.random.default_rng(0)
x = (rng.standard_normal(n) + 10).astype(np.float32)
g = pa.AccelSearchGPU(0)
g.load_dat(x, deredden=True)
t = time.perf_counter()
for _ in range(5):
    g.load_dat(x, deredden=True)
dt = (time.perf_counter() - t) / 5
print("n=%d: %.2f ms per load_dat (upload + realfft + deredden)" % (n, dt * 1e3))
t = time.perf_counter()
for _ in range(5):
    g.deredden()
print("deredden alone: %.2f ms" % ((time.perf_counter() - t) / 5 * 1e3))
