"""Random search geometries through every mode of the harmonic sum (hs_bound 0 = float plane, 1 = the library's choice,
3 = bounded sum alone, 4 = coarse bound forced): the hits must be the same bytes.  python scripts/fuzz_modes.py [n] [seed]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import presto_amd as pa                      # noqa: E402
from presto_amd import synth                 # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
g = pa.AccelSearchGPU(0)
bad = 0
for it in range(n):
    log2n = int(rng.integers(18, 23))
    zmax = int(rng.choice([0, 2, 10, 50, 100, 200, 300, 302, 400, 600, 1200]))
    if zmax >= 600:
        log2n = min(log2n, 21)
    nh = int(rng.choice([1, 2, 4, 8, 16, 32]))
    sigma = float(rng.choice([1.5, 2.0, 3.0]))
    N, dt = 1 << log2n, 1e-4
    fmax = 0.5 / dt
    flo = float(rng.choice([1.0, 13.0, 97.0]))
    fhi = float(rng.choice([0.4, 0.8, 0.98])) * fmax
    pulsars = [dict(f=float(rng.uniform(flo + 5, fhi / 2)), z=float(rng.uniform(-zmax, zmax)) * 0.4, amp=float(rng.uniform(0.05, 0.4)),
                    duty=float(rng.uniform(0.03, 0.2))) for _ in range(int(rng.integers(0, 3)))]
    fft = synth.make_fft(N, dt, pulsars, seed=int(rng.integers(1 << 30)))
    res = {}
    for mode in (0, 1, 3, 4):
        g.set_tuning("hs_bound", mode)
        g.configure(N // 2, N, dt, zmax, nh, sigma=sigma, flo=flo, fhi=fhi)
        g.load_fft(fft)
        res[mode] = g.run(max_hits=1 << 24)
    same = all(res[0].tobytes() == h.tobytes() for h in res.values())
    bad += not same
    print("%2d: 2^%d points zmax %4d numharm %2d sigma %.1f flo %.0f fhi %.0f, %d pulsars: %7d hits, coarse bound %s, %s" % (
        it, log2n, zmax, nh, sigma, flo, fhi, len(pulsars), len(res[0]), "ran" if g.get_tuning("hs_prefilter_active") else "n/a",
        "same" if same else "DIFFERENT " + str({m: len(h) for m, h in res.items()})), flush=True)
g.set_tuning("hs_bound", 1)
g.close()
print("%d of %d differ" % (bad, n))
sys.exit(1 if bad else 0)
