"""GPU parity tests of the standard (not in-memory) search mode against the oracle.

Bars as in test_gpu_parity.py: block powers relative to the block's largest value
(float32 FFTs with different butterfly orders; sub-harmonics whose reference
transform is 512 points run zero-padded at 1024 on the GPU), look-up tables
bit-exact, add_ffdotpows bit-exact on shared inputs, hit lists identical up to
cells within tolerance of a threshold.
"""
import numpy as np
import pytest

from presto_amd import synth

pytestmark = pytest.mark.gpu

USELEN = 7470
TOL = 2e-5


@pytest.fixture(scope="module")
def gpu():
    import presto_amd as pa
    g = pa.AccelSearchGPU(0)
    yield g
    g.close()


def _both(gpu, oracle, N, zmax, numharm, name="cfg2", **kw):
    fft, c = synth.config_fft(name, N=N)
    args = dict(sigma=2.0, flo=1.0, fhi=1e9)
    args.update(kw)
    s = oracle.Search(fft, c["N"], c["dt"], zmax, numharm, alloc_plane=False, nthreads=8, **args)
    gpu.configure(fft.size, c["N"], c["dt"], zmax, numharm, inmem=False, **args)
    gpu.load_fft(fft)
    return s, fft


HARMS = [(1, 1), (2, 1), (4, 1), (4, 3), (8, 3), (8, 7), (16, 1), (16, 5), (16, 15)]


@pytest.mark.parametrize("zmax,numharm", [(200, 16), (0, 8), (50, 32)])
def test_subharm_blocks(gpu, oracle, zmax, numharm):
    s, fft = _both(gpu, oracle, 1 << 18, zmax, numharm)
    harms = [h for h in HARMS if h[0] <= numharm]
    if numharm == 32:
        harms += [(32, 1), (32, 17), (32, 31)]
    for startr in (s.rlo, s.rlo + 3 * 3735.0):
        for nh, h in harms:
            want, rlo_w, zlo_w, rinds_w = s.subharm_block(nh, h, startr)
            got, rlo_g, zlo_g, rinds_g = gpu.subharm_block(nh, h, startr)
            assert (rlo_g, zlo_g) == (rlo_w, zlo_w), (nh, h)
            assert got.shape == want.shape, (nh, h, got.shape, want.shape)
            if nh > 1:
                assert np.array_equal(rinds_g, rinds_w), (nh, h)
            scale = want.max()
            assert np.abs(got - want).max() <= TOL * scale, (nh, h, np.abs(got - want).max() / scale)


def test_add_ffdotpows_bit_exact(gpu, oracle):
    s, fft = _both(gpu, oracle, 1 << 18, 200, 16)
    startr = s.rlo + 3735.0
    fund, _, _, _ = s.subharm_block(1, 1, startr)
    for nh, h in [(2, 1), (4, 3), (16, 7)]:
        sub, _, zlo, rinds = s.subharm_block(nh, h, startr)
        want = fund.copy()
        s.add_ffdotpows(want, sub, zlo, rinds, nh, h)
        got = fund.copy()
        gpu.add_ffdotpows(got, sub, zlo, rinds, nh, h)
        assert np.array_equal(got, want), (nh, h)


def _hit_key(h, stage_field):
    return (int(h["block"]), int(h[stage_field]), int(h["zind"]), int(h["rind"]))


@pytest.mark.parametrize("N,zmax,numharm,kw", [
    (1 << 19, 200, 16, {}),
    (1 << 18, 50, 32, {}),
    (1 << 18, 20, 8, dict(norm_type=1)),
])
def test_run_standard_matches_oracle(gpu, oracle, N, zmax, numharm, kw):
    gkw = dict(kw)
    okw = dict(kw)
    fft, c = synth.config_fft("cfg2", N=N)
    s = oracle.Search(fft, c["N"], c["dt"], zmax, numharm, alloc_plane=False, nthreads=8,
                      sigma=2.0, flo=1.0, fhi=1e9, **okw)
    gpu.configure(fft.size, c["N"], c["dt"], zmax, numharm, inmem=False, sigma=2.0, flo=1.0,
                  fhi=1e9, **gkw)
    gpu.load_fft(fft)
    ohits, ocands = s.run_standard()
    ghits, gcands, gsig, gadded = gpu.search_standard()
    assert len(ohits) > 0
    powcut = s.powcut
    ok = {(int(h["block"]), int(np.log2(h["numharm"])), int(h["zind"]), int(h["rind"])): h for h in ohits}
    gk = {_hit_key(h, "stage"): h for h in ghits}
    for k in set(ok) ^ set(gk):
        h = ok.get(k, gk.get(k))
        cut = powcut[k[1]]
        assert abs(float(h["power"]) - cut) <= 1e-4 * cut, ("not a threshold-edge cell", k, h)
    common = sorted(set(ok) & set(gk))
    assert len(common) >= 0.98 * max(len(ok), len(gk))
    po = np.array([ok[k]["power"] for k in common])
    pg = np.array([gk[k]["power"] for k in common])
    assert np.abs(po - pg).max() <= 1e-3 * po.max()
    if set(ok) == set(gk):
        assert len(gcands) == len(ocands)
        assert np.array_equal(gcands["r"], ocands["r"]) and np.array_equal(gcands["z"], ocands["z"])
        assert np.array_equal(gcands["numharm"], ocands["numharm"])
        assert np.allclose(gcands["sigma"], ocands["sigma"], rtol=1e-3)


def test_standard_mode_has_no_plane(gpu, oracle):
    import presto_amd as pa
    s, fft = _both(gpu, oracle, 1 << 18, 20, 2)
    with pytest.raises(pa.AccelGPUError):
        gpu.fill_plane()
    with pytest.raises(pa.AccelGPUError):
        gpu.get_plane(0, 0, 16)
    # back to in-memory mode on the same context
    gpu.configure(fft.size, 1 << 18, 64e-6, 20, 2, fhi=1e9)
    gpu.load_fft(fft)
    assert len(gpu.run()) >= 0
