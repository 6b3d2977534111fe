import numpy as np, sys
sys.path.insert(0, '.')
import presto_amd as pa
from presto_amd import synth
N, dt, zmax, numharm = 1 << 17, 1e-3, 20, 8
fft = synth.make_fft(N, dt, [dict(f=37.3, z=7.0, amp=0.35, duty=0.05)], seed=5)
g = pa.AccelSearchGPU(0)
su = g.configure(fft.size, N, dt, zmax, numharm, fhi=1e9)
g.load_fft(fft)
g.fill_plane()
startr = su.rlo + 3735.0
a, rlo = g.subharm_ffdot_plane(startr)
b, rlo2, zlo, _ = g.subharm_block(1, 1, startr)
p = np.stack([g.get_plane(z, 2 * rlo, 7470) for z in range(su.numz)])
print("fund_block vs subharm_block", np.array_equal(a, b), np.abs(a - b).max())
print("fund_block vs plane", np.array_equal(a, p), np.abs(a - p).max())
