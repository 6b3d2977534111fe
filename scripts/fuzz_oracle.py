"""Random small searches on the GPU (default path, through the C ABI) against the CPU oracle: hits matched by
(stage, z row, column), powers within float32 round-off of two FFT pipelines, one-sided hits only at a threshold.
A checker like the tests (it imports oracle/): python scripts/fuzz_oracle.py [n] [seed]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import presto_amd as pa                      # noqa: E402
from presto_amd import synth                 # noqa: E402
from oracle import binding                   # noqa: E402
from oracle.compare import compare_lists, PLANE_TOL, EDGE_TOL  # noqa: E402

binding.build()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
g = pa.AccelSearchGPU(0)
bad = 0
for it in range(n):
    zmax = int(rng.choice([0, 2, 20, 50, 100, 200, 300, 302, 400, 600]))
    log2n = int(rng.integers(16, 20)) if zmax < 300 else int(rng.integers(16, 18))
    nh = int(rng.choice([1, 2, 4, 8, 16, 32]))
    sigma = float(rng.choice([2.0, 3.0, 4.0]))
    N, dt = 1 << log2n, 1e-4
    flo = float(rng.choice([1.0, 13.0, 97.0]))
    fhi = float(rng.choice([0.4, 0.8, 0.98])) * 0.5 / dt
    pulsars = [dict(f=float(rng.uniform(flo + 5, fhi / 2)), z=float(rng.uniform(-zmax, zmax)) * 0.4, amp=float(rng.uniform(0.1, 0.4)),
                    duty=float(rng.uniform(0.03, 0.2))) for _ in range(int(rng.integers(1, 3)))]
    fft = synth.make_fft(N, dt, pulsars, seed=int(rng.integers(1 << 30)))
    s = binding.Search(fft, N, dt, zmax, nh, nthreads=16, sigma=sigma, flo=flo, fhi=fhi)
    want_hits, want_cands = s.run()
    setup = g.configure(N // 2, N, dt, zmax, nh, sigma=sigma, flo=flo, fhi=fhi)
    g.load_fft(fft)
    got_hits, got_cands, sig, added = g.search()
    rel, edge_g, edge_w = compare_lists(setup.powcut, got_hits, want_hits)
    ok = len(rel) >= 0.98 * len(want_hits) and (len(rel) == 0 or rel.max() <= 10 * PLANE_TOL) \
        and (edge_g <= EDGE_TOL).all() and (edge_w <= EDGE_TOL).all()
    same_cands = len(got_cands) == len(want_cands)
    bad += not ok
    print("%2d: 2^%d points zmax %3d numharm %2d sigma %.0f flo %.0f fhi %.0f: oracle %6d hits, matched %6d, power diff <= %.1e, "
          "one-sided %d + %d (all at a threshold: %s), candidates %d / %d %s" % (
              it, log2n, zmax, nh, sigma, flo, fhi, len(want_hits), len(rel), rel.max() if len(rel) else 0.0, len(edge_g), len(edge_w),
              bool((edge_g <= EDGE_TOL).all() and (edge_w <= EDGE_TOL).all()), len(got_cands), len(want_cands),
              "ok" if ok else "MISMATCH"), flush=True)
g.close()
print("%d of %d outside the bars" % (bad, n))
sys.exit(1 if bad else 0)
