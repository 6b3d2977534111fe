"""Regenerates tests/golden/search_small.npz: a small synthetic spectrum and the
CPU oracle's outputs for it (hits in visit order with sigma and *added flags,
final candidate list, thresholds).  The reference program cannot be built in
this image (it needs FFTW), so these vectors come from the oracle, whose pieces
are pinned separately (tests/test_oracle_pins.py); they freeze its behaviour.
Run from the repo root: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob          # noqa: E402
from presto_amd import synth             # noqa: E402

N, dt, zmax, numharm = 1 << 16, 2e-3, 30, 8
fft = synth.make_fft(N, dt, [dict(f=23.7, z=11.0, amp=0.5, duty=0.06),
                             dict(f=91.13, z=-24.0, amp=0.6, duty=0.04)], seed=2024)
s = ob.Search(fft, N, dt, zmax, numharm, fhi=1e9)
hits, cands = s.run()
blk, _ = s.fund_block(s.rlo)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "search_small.npz"),
                    fft=fft, N=N, dt=dt, zmax=zmax, numharm=numharm,
                    powcut=s.powcut, numindep=s.numindep, rlo=s.rlo, highestbin=s.highestbin,
                    hits=hits, cands=cands, block0_rows=blk[::5, ::37].copy(),
                    kernel_z10=ob.init_kernel(10, s.fftlen)[0][::16].copy())
print("hits", len(hits), "cands", len(cands), "added", int(hits["added"].sum()))

# ---- the tail and the pre-steps: tests/golden/post_small.npz + post_small_ACCEL_30 -------
opt = s.optimize(cands, polish=True, eliminate=True)
inf = synth.inf_text("post_small_DM7.50", N, dt)
accel = os.path.join(ROOT, "tests", "golden", "post_small_DM7.50_ACCEL_30")
s.write_accel_files(opt, accel, accel + ".cand", inf)
rng = np.random.default_rng(11)
series = (rng.standard_normal(1 << 13) + 0.05 * np.cumsum(rng.standard_normal(1 << 13)) + 5).astype(np.float32)
spec = ob.realfft(series)
birds_lo, birds_hi = np.array([700.2, 1500.0]), np.array([712.9, 1501.0])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "post_small.npz"),
                    opt=opt, series=series, realfft=spec, deredden=ob.deredden(spec),
                    rednoise=ob.rednoise(spec, (1 << 13) * 1e-3), birds_lo=birds_lo, birds_hi=birds_hi,
                    zapped=ob.zapbirds(spec, birds_lo, birds_hi))
print("optimised", len(opt), "of", len(cands))
