import numpy as np, sys
sys.path.insert(0, '.')
import presto_amd as pa
from oracle import binding as ob
rng = np.random.default_rng(3)
n = 1 << 12
x = (rng.standard_normal(1 << 16) + 0.05 * np.cumsum(rng.standard_normal(1 << 16)) + 10).astype(np.float32)
spec = ob.realfft(x)[:n].copy()
want = ob.deredden(spec)
g = pa.AccelSearchGPU(0)
g.load_fft(spec); g.deredden(); got = g.get_fft(n)
bad = np.nonzero(got.view(np.uint32) != want.view(np.uint32))[0]
print("mismatching floats", bad.size, "of", 2 * n, "first", bad[:20] // 2)
d = np.abs(got - want) / np.abs(want)
print("max rel", d.max(), "at", d.argmax())
i = bad[0] // 2 if bad.size else 0
print(got[i], want[i], spec[i], (got[i] / spec[i]), (want[i] / spec[i]))
