"""GPU parity tests: time-series input (realfft + deredden) and the rednoise filter
against the oracle.

Bars: the long FFT is float32 on both sides with different factorizations ->
compared relative to the spectrum's rms amplitude times sqrt(log2 n) (5e-7 per
stage); the red-noise filters are compared on ONE shared spectrum, where block
medians are bit-exact and every scaled amplitude is bit-exact (un-fused float
arithmetic in the reference's order).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import presto_amd as pa
    g = pa.AccelSearchGPU(0)
    yield g
    g.close()


def _series(n, seed, red=0.0):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(n)
    if red:
        x = x + red * np.cumsum(rng.standard_normal(n))
    t = np.arange(n)
    x += 0.3 * (np.modf(t * 0.01234567)[0] < 0.05)
    return (x + 10.0).astype(np.float32)


@pytest.mark.parametrize("log2n", [11, 13, 15, 16, 17, 18, 19, 20, 22])
def test_realfft_matches_oracle(gpu, oracle, log2n):
    n = 1 << log2n
    x = _series(n, log2n)
    want = oracle.realfft(x)
    assert gpu.load_dat(x, deredden=False) == n // 2
    got = gpu.get_fft(n // 2)
    # DC and Nyquist are sums of n terms of size ~10: compare them relative to themselves
    assert abs(got[0].real - want[0].real) <= 2e-6 * abs(want[0].real)
    rms = np.sqrt(np.mean(np.abs(want[1:]) ** 2))
    # float32 round-off scales with the largest term in the butterflies: the DC sum
    tol = 3e-6 * rms * np.sqrt(log2n) + 2e-7 * abs(want[0].real)
    err = np.abs(got[1:] - want[1:]).max()
    assert err <= tol, (log2n, err / rms)
    # and against a double-precision transform
    ref = np.fft.rfft(x.astype(np.float64))[1:-1]
    assert np.abs(got[1:] - ref).max() <= tol


def test_realfft_rejects_bad_lengths(gpu):
    import presto_amd as pa
    with pytest.raises(pa.AccelGPUError):
        gpu.load_dat(np.zeros(4097, dtype=np.float32))          # odd
    with pytest.raises(pa.AccelGPUError):
        gpu.load_dat(np.zeros(512, dtype=np.float32))           # too short


@pytest.mark.parametrize("n", [3000, 2 * 3 * 5 * 7 * 11 * 4, 100002, 3 * (1 << 17), 1000000, 5 * (1 << 20) + 2])
def test_realfft_any_even_length(gpu, oracle, n):
    """Lengths that are not powers of two go through the chirp-z convolution."""
    x = _series(n, n % 97)
    assert gpu.load_dat(x, deredden=False) == n // 2
    got = gpu.get_fft(n // 2)
    ref = np.fft.rfft(x.astype(np.float64))
    rms = np.sqrt(np.mean(np.abs(ref[1:-1]) ** 2))
    tol = 4e-6 * rms * np.sqrt(np.log2(n)) + 4e-7 * abs(ref[0].real)
    assert abs(got[0].real - ref[0].real) <= 4e-6 * abs(ref[0].real)
    assert np.abs(got[1:] - ref[1:-1]).max() <= tol
    if n <= 200000:
        want = oracle.realfft(x)
        assert np.abs(got[1:] - want[1:]).max() <= tol


@pytest.mark.parametrize("n", [1 << 12, 1 << 16, 1 << 20, (1 << 18) + 1234])
def test_deredden_bit_exact_on_shared_spectrum(gpu, oracle, n):
    spec = oracle.realfft(_series(1 << 21, 3, red=0.05))[:n].copy()
    want = oracle.deredden(spec)
    gpu.load_fft(spec)
    gpu.deredden()
    got = gpu.get_fft(n)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert got[0] == 1.0


@pytest.mark.parametrize("n,kw", [(1 << 16, {}), (1 << 20, {}), ((1 << 17) + 77, dict(startwidth=4, endwidth=64, endfreq=2.0)),
                                  (1 << 14, dict(startwidth=10, endwidth=300, endfreq=0.5))])
def test_rednoise_bit_exact_on_shared_spectrum(gpu, oracle, n, kw):
    spec = oracle.realfft(_series(1 << 21, 5, red=0.1))[:n].copy()
    T = 2 * n * 1e-3
    want = oracle.rednoise(spec, T, **kw)
    gpu.load_fft(spec)
    gpu.rednoise(T, **kw)
    got = gpu.get_fft(n)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # the program's last half block: imaginary parts are copies of the real parts
    assert np.array_equal(got[-3:].real, got[-3:].imag)


def test_dat_input_end_to_end(gpu, oracle):
    """.dat path: series -> realfft -> deredden -> search equals the oracle doing the same."""
    n = 1 << 19
    x = _series(n, 9, red=0.02)
    spec = oracle.deredden(oracle.realfft(x))
    spec[0] = 1.0 + 1.0j          # time-series input without -photon (src/accel_utils.c:1553-1556)
    s = oracle.Search(spec, n, 1e-3, 20, 8, fhi=1e9, nthreads=8)
    ohits, ocands = s.run()
    gpu.configure(n // 2, n, 1e-3, 20, 8, fhi=1e9)
    gpu.load_dat(x, deredden=True)
    gpu.set_bin0(1.0, 1.0)
    assert gpu.get_fft(n // 2)[0] == np.complex64(1.0 + 1.0j)
    # the pre-population block at r = 8 has bin 0 inside its window
    want0, _ = s.fund_block(8.0)
    got0, _ = gpu.subharm_ffdot_plane(8.0)
    assert np.abs(got0 - want0).max() <= 1e-3 * want0.max()
    ghits, gcands, sig, added = gpu.search()
    assert len(ocands) > 0
    ok = {(int(h["block"]), int(np.log2(h["numharm"])), int(h["zind"]), int(h["rind"])): float(h["power"]) for h in ohits}
    gk = {(int(h["block"]), int(h["stage"]), int(h["zind"]), int(h["rind"])): float(h["power"]) for h in ghits}
    common = set(ok) & set(gk)
    assert len(common) >= 0.97 * max(len(ok), len(gk))
    for k in set(ok) ^ set(gk):
        p = ok.get(k, gk.get(k))
        cut = s.powcut[k[1]]
        assert abs(p - cut) <= 2e-3 * cut, k
    po = np.array([ok[k] for k in common]); pg = np.array([gk[k] for k in common])
    assert np.abs(po - pg).max() <= 1e-3 * po.max()
    best_o = ocands[np.argmax(ocands["sigma"])]; best_g = gcands[np.argmax(gcands["sigma"])]
    assert (best_o["r"], best_o["z"], best_o["numharm"]) == (best_g["r"], best_g["z"], best_g["numharm"])


def test_zapbirds_bit_exact_on_shared_spectrum(gpu, oracle):
    n = 1 << 18
    spec = oracle.realfft(_series(1 << 19, 8, red=0.02)).copy()
    # overlapping and adjacent ranges (a later bird's median window covers an earlier
    # zap), one too low to be zapped, one hanging over the top, one past it
    lo = np.array([120.0, 500.25, 760.5, 9000.0, 9150.7, 60000.0, n - 50.5, n + 10.0])
    hi = np.array([130.0, 520.75, 790.1, 9100.3, 9151.7, 60400.0, n + 20.0, n + 30.0])
    want = oracle.zapbirds(spec, lo, hi)
    gpu.load_fft(spec)
    gpu.zapbirds(lo, hi)
    got = gpu.get_fft(n)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(got[100:140], spec[100:140])                  # below bin 301: untouched
    assert got[510].imag == 0.0 and got[510].real > 0 and got[510] == got[505]
    assert not np.array_equal(got[60000:60400], spec[60000:60400])


def test_golden_pre_steps(gpu):
    """Committed vectors (tests/golden/post_small.npz): the three in-place filters reproduce
    them bit for bit from the golden spectrum; the device transform of the golden series
    agrees with the golden spectrum to float round-off."""
    import os
    p = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "post_small.npz"))
    spec = p["realfft"]
    n = spec.size
    gpu.load_fft(spec); gpu.deredden()
    assert np.array_equal(gpu.get_fft(n).view(np.uint32), p["deredden"].view(np.uint32))
    gpu.load_fft(spec); gpu.rednoise(2 * n * 1e-3)
    assert np.array_equal(gpu.get_fft(n).view(np.uint32), p["rednoise"].view(np.uint32))
    gpu.load_fft(spec); gpu.zapbirds(p["birds_lo"], p["birds_hi"])
    assert np.array_equal(gpu.get_fft(n).view(np.uint32), p["zapped"].view(np.uint32))
    gpu.load_dat(p["series"], deredden=False)
    got = gpu.get_fft(n)
    rms = np.sqrt(np.mean(np.abs(spec[1:]) ** 2))
    assert np.abs(got[1:] - spec[1:]).max() <= 1e-5 * rms + 2e-7 * abs(spec[0].real)
