"""GPU parity tests: the HIP library (through its C ABI) against the CPU oracle.

Bars:
  * kernel bank, fundamental blocks, plane: float32 FFT pipelines with different
    butterfly orders on the two sides -> compared relative to the largest value
    in the compared array, tolerance 2e-5 (north-star bar for powers is 1e-3).
  * harmonic sum + threshold on ONE shared plane: bit-exact (float adds in the
    reference's order, integer index arithmetic).
  * end-to-end candidate lists: identical (r, z, numharm) bins; power and sigma
    within 1e-3 relative; cells within 2e-5 of a threshold are classified as
    threshold-edge cases and may appear on one side only.
"""
import numpy as np
import pytest

from presto_amd import synth

pytestmark = pytest.mark.gpu

USELEN = 7470
PLANE_TOL = 2e-5


@pytest.fixture(scope="module")
def gpu():
    import presto_amd as pa
    g = pa.AccelSearchGPU(0)
    yield g
    g.close()


def _case(name, N, zmax=None, numharm=None, seed=None):
    fft, c = synth.config_fft(name, N=N, seed=seed)
    if zmax is not None:
        c["zmax"] = zmax
    if numharm is not None:
        c["numharm"] = numharm
    return fft, c


def _both(gpu, oracle, fft, c, **kw):
    args = dict(sigma=2.0, flo=1.0, fhi=1e9)
    args.update(kw)
    s = oracle.Search(fft, c["N"], c["dt"], c["zmax"], c["numharm"], nthreads=8, **args)
    setup = gpu.configure(fft.size, c["N"], c["dt"], c["zmax"], c["numharm"], **args)
    gpu.load_fft(fft)
    return s, setup


def test_setup_matches_oracle(gpu, oracle):
    for zmax, numharm, N in ((0, 8, 1 << 20), (200, 16, 1 << 20), (300, 16, 1 << 21), (1200, 32, 1 << 20)):
        fft = np.zeros(N // 2, dtype=np.complex64)
        c = dict(N=N, dt=64e-6, zmax=zmax, numharm=numharm)
        s = oracle.Search(fft, N, c["dt"], zmax, numharm, fhi=1e9, alloc_plane=False)
        import presto_amd as pa
        p = pa.plan(fft.size, N, c["dt"], zmax, numharm, fhi=1e9)
        assert (p.rlo, p.highestbin, p.numz, p.fftlen, p.numharmstages) == \
            (s.rlo, s.highestbin, s.numz, s.fftlen, s.numstages)
        assert p.num_search_blocks == s.num_blocks and p.num_prepop_blocks == s.num_prepop
        assert list(p.numindep)[:p.numharmstages] == list(s.numindep)
        assert np.array_equal(np.array(list(p.powcut)[:p.numharmstages], dtype=np.float32), s.powcut)
        assert p.rlen == s.rlen


@pytest.mark.parametrize("zmax", [0, 50, 200, 300, 1200])
def test_kernel_bank(gpu, oracle, zmax):
    gpu.create_subharminfos(zmax)
    L = oracle.lib()
    fftlen = L.orc_calc_fftlen(1, 1, zmax)
    numz = (zmax // 2) * 2 + 1
    picks = sorted(set([0, 1, numz // 4, numz // 2, numz - 2, numz - 1]) & set(range(numz)))
    for zi in picks:
        want, _ = oracle.init_kernel(-zmax + 2 * zi, fftlen)
        got = gpu.kernel(zi, fftlen)
        scale = np.abs(want).max()
        assert np.abs(got - want).max() <= 4e-6 * scale, (zmax, zi)


@pytest.mark.parametrize("name,N,zmax,numharm", [("cfg1", 1 << 20, 0, 8), ("cfg2", 1 << 19, 200, 16),
                                                 ("cfg2", 1 << 18, 300, 16), ("cfg5", 1 << 19, 1200, 32)])
def test_fundamental_blocks(gpu, oracle, name, N, zmax, numharm):
    fft, c = _case(name, N, zmax, numharm)
    s, setup = _both(gpu, oracle, fft, c)
    starts = [8.0, s.rlo, s.rlo + 3735.0 * (s.num_blocks - 1)]
    for startr in starts:
        want, rlo_w = s.fund_block(startr)
        got, rlo_g = gpu.subharm_ffdot_plane(startr)
        assert rlo_w == rlo_g
        scale = want.max()
        err = np.abs(got - want).max() / scale
        assert err <= PLANE_TOL, (startr, err)
    s.close()


def test_ragged_ends_and_lobin(gpu, oracle):
    """Windows that run off both ends of the file (zero padding) and a file that
    starts at a non-zero bin (-lobin)."""
    fft, c = _case("cfg2", 1 << 17, 200, 4)
    s, setup = _both(gpu, oracle, fft, c)
    for startr in (8.0, float(fft.size - 3737), float(fft.size - 3736 - 60)):
        want, _ = s.fund_block(startr)
        got, _ = gpu.subharm_ffdot_plane(startr)
        assert np.abs(got - want).max() <= PLANE_TOL * want.max()
    s.close()
    s, setup = _both(gpu, oracle, fft, c, lobin=1000, flo=0.0)
    assert setup.rlo == 1000
    for startr in (1000.0, 1000.0 + 3735.0):
        want, _ = s.fund_block(startr)
        got, _ = gpu.subharm_ffdot_plane(startr)
        assert np.abs(got - want).max() <= PLANE_TOL * want.max()
    s.close()


def test_photon_normalisation(gpu, oracle):
    fft, c = _case("cfg1", 1 << 18, 20, 2)
    fft = fft.copy()
    fft[0] = np.complex64(complex(7.5, 0.3))       # "number of photons" lives in bin 0
    s = oracle.Search(fft, c["N"], c["dt"], 20, 2, fhi=1e9, photon=True)
    setup = gpu.configure(fft.size, c["N"], c["dt"], 20, 2, fhi=1e9, photon_nph=float(fft[0].real))
    gpu.load_fft(fft)
    want, _ = s.fund_block(s.rlo)
    got, _ = gpu.subharm_ffdot_plane(s.rlo)
    assert np.abs(got - want).max() <= PLANE_TOL * want.max()
    s.close()


@pytest.mark.parametrize("name,N,zmax,numharm", [("cfg1", 1 << 19, 0, 8), ("cfg2", 1 << 18, 200, 16),
                                                 ("cfg2", 1 << 18, 40, 32), ("cfg2", 1 << 18, 60, 4),
                                                 ("cfg2", 1 << 18, 60, 2), ("cfg2", 1 << 18, 60, 1)])
def test_harmonic_sum_and_threshold_bit_exact(gpu, oracle, name, N, zmax, numharm):
    """Both sides sum and threshold the SAME plane (the GPU's): every hit must
    match bit for bit, in the same order, and so must the candidate lists."""
    import presto_amd as pa
    fft, c = _case(name, N, zmax, numharm)
    s, setup = _both(gpu, oracle, fft, c)
    gpu.fill_plane()
    s.set_plane(gpu.plane())
    want_hits, want_cands = s.search_plane()
    got = gpu.search_plane()
    assert len(got) == len(want_hits) and len(got) > 0
    assert np.array_equal(got["power"].view(np.uint32), want_hits["power"].view(np.uint32))
    assert np.array_equal(1 << got["stage"], want_hits["numharm"])
    assert np.array_equal(got["zind"], want_hits["zind"])
    assert np.array_equal(got["rind"], want_hits["rind"])
    assert np.array_equal(got["block"], want_hits["block"])
    cands, sig, added = pa.api.replay_candidates(setup, got)
    assert np.array_equal(added, want_hits["added"])
    np.testing.assert_allclose(sig, want_hits["sigma"], rtol=1e-6)
    assert len(cands) == len(want_cands)
    assert np.array_equal(cands["r"], want_cands["r"]) and np.array_equal(cands["z"], want_cands["z"])
    assert np.array_equal(cands["numharm"], want_cands["numharm"])
    assert np.array_equal(cands["power"].view(np.uint32), want_cands["power"].view(np.uint32))
    # summed blocks after every stage, straight from the getter
    blk = min(1, setup.num_search_blocks - 1)
    rlo_b = int(s.rlo) + blk * 3735
    pw = s.plane()[:, 2 * rlo_b: 2 * rlo_b + USELEN].copy()   # a copy: the sums must not land in the plane
    assert np.array_equal(gpu.summed_block(blk, 0), pw)
    for stage in range(1, setup.numharmstages):
        for h in range(1, 1 << stage, 2):
            s.inmem_add(pw, rlo_b, 1 << stage, h)
        assert np.array_equal(gpu.summed_block(blk, stage).view(np.uint32), pw.view(np.uint32)), stage
    s.close()


def _compare_lists(setup, got_hits, want_hits):
    """Match hits by (stage, zind, rind); unmatched ones must sit at a threshold."""
    from oracle.compare import compare_lists
    return compare_lists(setup.powcut, got_hits, want_hits)


@pytest.mark.parametrize("name,N,zmax,numharm", [("cfg1", 1 << 20, 0, 8), ("cfg2", 1 << 19, 200, 16),
                                                 ("cfg5", 1 << 18, 1200, 32)])
def test_end_to_end_candidates(gpu, oracle, name, N, zmax, numharm):
    fft, c = _case(name, N, zmax, numharm)
    s, setup = _both(gpu, oracle, fft, c)
    want_hits, want_cands = s.run()
    got_hits, got_cands, sig, added = gpu.search()
    rel, edge_g, edge_w = _compare_lists(setup, got_hits, want_hits)
    assert len(rel) >= 0.98 * max(len(want_hits), 1)
    assert rel.max() <= 1e-3                      # north-star bar
    assert rel.max() <= 10 * PLANE_TOL            # what we actually reach
    assert (edge_g <= 4 * PLANE_TOL).all() and (edge_w <= 4 * PLANE_TOL).all()
    if len(edge_g) == 0 and len(edge_w) == 0:
        # no threshold-edge flips: the final lists must name the same bins
        assert len(got_cands) == len(want_cands)
        assert np.array_equal(got_cands["r"], want_cands["r"])
        assert np.array_equal(got_cands["z"], want_cands["z"])
        assert np.array_equal(got_cands["numharm"], want_cands["numharm"])
        np.testing.assert_allclose(got_cands["power"], want_cands["power"], rtol=1e-3)
        np.testing.assert_allclose(got_cands["sigma"], want_cands["sigma"], rtol=1e-3, atol=1e-3)
    # both sides rank the same candidate first
    best, wbest = got_cands[np.argmax(got_cands["sigma"])], want_cands[np.argmax(want_cands["sigma"])]
    assert best["r"] == wbest["r"] and best["z"] == wbest["z"] and best["numharm"] == wbest["numharm"]
    assert abs(best["sigma"] - wbest["sigma"]) <= 1e-3 * wbest["sigma"]
    s.close()


def test_injected_pulsar_recovered_at_known_rz(gpu):
    """Oracle-free property: a drifting tone injected at (r0, z0) is found in the
    matching z row at r0, with close to the injected power."""
    numbins = 1 << 18
    N = 2 * numbins
    r0, z0, amp = 50000.25, 80.0, 12.0
    fft = synth.noise_fft(numbins, 7, lines=[(r0, z0, amp)])
    setup = gpu.configure(numbins, N, 1e-3, 200, 1, flo=0.01, fhi=1e9)
    gpu.load_fft(fft)
    hits, cands, sig, added = gpu.search()
    best = cands[np.argmax(cands["power"])]
    assert abs(best["r"] - r0) <= 0.5 and abs(best["z"] - z0) <= 2.0
    assert 0.7 * amp * amp <= best["power"] <= 1.4 * amp * amp


def test_tiled_correlate_grid_equals_plain_grid(gpu, monkeypatch):
    """Plane generation maps workgroups to (template, chunk) pairs in tiles per XCD when
    the grid is large.  Any mapping must give the same plane: compare the default with
    the plain two-dimensional grid and with odd tile shapes, bit for bit (hits carry the
    summed powers; plane rows are read back directly)."""
    import os
    numbins = 1 << 21                       # 561 blocks -> 36 chunks x 201 templates: tiled by default
    fft = synth.noise_fft(numbins, 21, lines=[(300000.5, 64.0, 10.0), (1200000.25, -150.0, 9.0)])
    setup = gpu.configure(numbins, 2 * numbins, 64e-6, 200, 8, fhi=1e9)
    gpu.load_fft(fft)
    assert setup.numz * ((setup.num_search_blocks + setup.num_prepop_blocks + 15) // 16) >= 4096
    default_tz, default_tc = gpu.get_tuning("corr_tz"), gpu.get_tuning("corr_tc")
    ref_hits = gpu.run()
    rows = [(0, 0), (100, 2 * 300000 - 64), (setup.numz - 1, setup.rlen - 40000)]
    ref_rows = [gpu.get_plane(z, c, 30000).copy() for z, c in rows]
    assert len(ref_hits) > 0
    for tz, tc in ((0, 1), (1, 1), (3, 5), (7, 64), (64, 3)):
        gpu.set_tuning("corr_tz", tz)
        gpu.set_tuning("corr_tc", tc)
        assert (gpu.get_tuning("corr_tz"), gpu.get_tuning("corr_tc")) == (tz, tc)
        hits = gpu.run()
        assert np.array_equal(hits, ref_hits), (tz, tc)
        for (z, c), want in zip(rows, ref_rows):
            assert np.array_equal(gpu.get_plane(z, c, 30000), want), (tz, tc, z)
    gpu.set_tuning("corr_tz", default_tz)
    gpu.set_tuning("corr_tc", default_tc)


def test_async_launch_collect_equals_run(gpu):
    numbins = 1 << 17
    fft = synth.noise_fft(numbins, 11, lines=[(30000.0, -20.0, 9.0)])
    gpu.configure(numbins, 2 * numbins, 1e-3, 100, 8, flo=0.02, fhi=1e9)
    gpu.load_fft(fft)
    a = gpu.run()
    gpu.run_launch()
    b = gpu.run_collect()
    assert np.array_equal(a, b)
    # a tiny caller buffer forces the overflow path and still returns every hit
    c = gpu.run(max_hits=1)
    assert np.array_equal(a, c)
    # two launches in flight (double-buffered hits), collected in order
    fft2 = synth.noise_fft(numbins, 12, lines=[(41000.5, 30.0, 10.0)])
    gpu.load_fft(fft2)
    b2 = gpu.run()
    gpu.load_fft(fft)
    gpu.run_launch()
    gpu.load_fft(fft2)
    gpu.run_launch()
    with pytest.raises(Exception):
        gpu.run_launch()                   # a third launch must be refused
    assert np.array_equal(gpu.run_collect(), a)
    assert np.array_equal(gpu.run_collect(), b2)
    assert not np.array_equal(a, b2)


def test_errors_are_reported(gpu):
    import presto_amd as pa
    with pytest.raises(pa.AccelGPUError):
        pa.plan(1 << 16, 1 << 17, 1e-3, 200, 3)            # numharm not a power of two
    with pytest.raises(pa.AccelGPUError):
        pa.plan(1 << 16, 1 << 17, 1e-3, 200, 8, flo=1e6)   # flo above the spectrum
    g = pa.AccelSearchGPU(0)
    with pytest.raises(pa.AccelGPUError):
        g.fill_plane()                                      # nothing configured
    g.close()


def test_golden_fixture(gpu):
    """The committed golden vectors (tests/golden/search_small.npz): inputs plus the
    oracle's hits, flags and candidates, checked without running the oracle."""
    import os
    import presto_amd as pa
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "search_small.npz"))
    setup = gpu.configure(g["fft"].size, int(g["N"]), float(g["dt"]), int(g["zmax"]), int(g["numharm"]), fhi=1e9)
    assert np.array_equal(np.array(list(setup.powcut)[:setup.numharmstages], dtype=np.float32), g["powcut"])
    gpu.load_fft(g["fft"])
    hits, cands, sig, added = gpu.search()
    rel, edge_g, edge_w = _compare_lists(setup, hits, g["hits"])
    assert len(rel) >= 0.99 * len(g["hits"]) and rel.max() <= 10 * PLANE_TOL
    assert (edge_g <= 4 * PLANE_TOL).all() and (edge_w <= 4 * PLANE_TOL).all()
    if len(edge_g) == 0 and len(edge_w) == 0:
        assert np.array_equal(added, g["hits"]["added"])
        assert np.array_equal(cands["r"], g["cands"]["r"]) and np.array_equal(cands["z"], g["cands"]["z"])
        assert np.array_equal(cands["numharm"], g["cands"]["numharm"])
    blk, rlo = gpu.subharm_ffdot_plane(float(g["rlo"]))
    assert np.abs(blk[::5, ::37] - g["block0_rows"]).max() <= PLANE_TOL * g["block0_rows"].max()
    k = gpu.kernel((int(g["zmax"]) + 10) // 2, setup.fftlen)
    assert np.abs(k[::16] - g["kernel_z10"]).max() <= 4e-6 * np.abs(g["kernel_z10"]).max()


@pytest.mark.parametrize("hs_bound", [1, 4])
@pytest.mark.parametrize("by_cost", [False, True])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_block_range_split_equals_whole_search(gpu, world, by_cost, hs_bound):
    """One long series split into `world` block ranges -- of equal length, or of equal predicted work (uneven) --
    each regenerating the plane rows its look-ups need and no others, gives, after merging, exactly the hits of
    one search.  hs_bound = 4: with the coarse bound (group maxima written for the generated rows alone)."""
    import presto_amd as pa
    from presto_amd import batch
    numbins = 1 << 18
    fft = synth.noise_fft(numbins, 31, lines=[(20000.5, 12.0, 11.0), (150001.0, -40.0, 12.0), (4000.0, 0.0, 9.0)])
    setup = gpu.configure(numbins, 2 * numbins, 1e-3, 60, 16, flo=0.3, fhi=1e9)
    gpu.load_fft(fft)
    gpu.set_tuning("hs_bound", hs_bound)
    try:
        _block_range_split(gpu, setup, world, by_cost, batch)
    finally:
        gpu.set_tuning("hs_bound", 1)


def _block_range_split(gpu, setup, world, by_cost, batch):
    from presto_amd.api import block_range_need
    whole = gpu.run()
    total_blocks = setup.num_search_blocks + setup.num_prepop_blocks
    parts, generated = [], []
    ranges = batch.split_blocks_by_cost(setup, world) if by_cost else batch.split_blocks(setup.num_search_blocks, world)
    assert ranges[0][0] == 0 and ranges[-1][1] == setup.num_search_blocks
    assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    for rank in range(world):
        first, last = ranges[rank]
        assert block_range_need(setup, first, last - first)[0] <= total_blocks
        # scribble over the plane so that nothing survives from the previous range
        for z in (0, setup.numz // 2, setup.numz - 1):
            gpu.put_plane(z, 0, np.full(setup.rlen, 1e9, dtype=np.float32))
        generated.append(gpu.set_block_range(first, last - first))
        parts.append(gpu.run())
        if len(parts[-1]):
            assert parts[-1]["block"].min() >= first and parts[-1]["block"].max() < last
    merged = batch.reference_order(np.concatenate(parts))
    assert np.array_equal(merged, whole)
    assert max(generated) <= total_blocks and generated[0] < total_blocks
    # round trip through the record form used on the wire
    assert np.array_equal(batch.records_to_hits(batch.hits_to_records(whole), whole.dtype), whole)
    gpu.set_block_range(0, setup.num_search_blocks)
    assert np.array_equal(gpu.run(), whole)


def test_locpow_normalisation(gpu, oracle):
    """-locpow: local-power normalisation instead of the block median."""
    fft, c = _case("cfg2", 1 << 18, 100, 8)
    s = oracle.Search(fft, c["N"], c["dt"], 100, 8, fhi=1e9, norm_type=1, nthreads=8)
    setup = gpu.configure(fft.size, c["N"], c["dt"], 100, 8, fhi=1e9, norm_type=1)
    gpu.load_fft(fft)
    for startr in (8.0, s.rlo, s.rlo + 3735.0 * (s.num_blocks - 1)):
        want, _ = s.fund_block(startr)
        got, _ = gpu.subharm_ffdot_plane(startr)
        assert np.abs(got - want).max() <= PLANE_TOL * want.max()
    want_hits, want_cands = s.run()
    got_hits, got_cands, sig, added = gpu.search()
    rel, edge_g, edge_w = _compare_lists(setup, got_hits, want_hits)
    assert len(rel) >= 0.98 * len(want_hits) and rel.max() <= 10 * PLANE_TOL
    assert (edge_g <= 4 * PLANE_TOL).all() and (edge_w <= 4 * PLANE_TOL).all()
    s.close()


def test_series_too_short_for_one_block(gpu, oracle):
    """The main loop runs while startr + 3735 < highestbin (src/accelsearch.c:168): a
    spectrum shorter than one block is searched by nobody -- no blocks, no hits, no
    candidates, no error; the oracle agrees; the pipelined calls work all the same."""
    N, dt = 6000, 1e-3
    rng = np.random.default_rng(3)
    fft = (rng.standard_normal(N // 2) + 1j * rng.standard_normal(N // 2)).astype(np.complex64)
    s = oracle.Search(fft, N, dt, 20, 8, fhi=1e9, nthreads=2)
    whits, wcands = s.run()
    s.close()
    setup = gpu.configure(fft.size, N, dt, 20, 8, fhi=1e9)
    assert setup.num_search_blocks == 0 == s.num_blocks and len(whits) == 0 and len(wcands) == 0
    gpu.load_fft(fft)
    hits, cands, sig, added = gpu.search()
    assert len(hits) == 0 and len(cands) == 0
    gpu.run_launch()
    gpu.run_launch()
    assert len(gpu.run_collect()) == 0 and len(gpu.run_collect()) == 0
    assert len(gpu.search_plane()) == 0


@pytest.mark.parametrize("zmax,z,r", [(0, 0, 9000.0), (50, 30, 9000.5), (200, -200, 12000.0), (200, 126, 7777.5),
                                      (300, 300, 9000.0), (1200, -1100, 11000.5)])
def test_matched_template_recovers_the_tone_power(gpu, zmax, z, r):
    """Oracle-free known answer, the one of the reference's own demonstration
    (python/ffdot_example.py:8-26): a noiseless unit-amplitude tone drifting by z bins around
    mean frequency r puts N^2/4 into the F-Fdot cell (r, z) -- all but the few percent the
    truncated templates leave out -- and no cell of its block holds more; the search lists
    it first."""
    N = 1 << 16
    fft = synth.chirp_fft(N, r, z)
    fft[0] = 1.0
    setup = gpu.configure(fft.size, N, 1e-3, zmax, 1, flo=0.0, fhi=1e9, photon_nph=1.0)
    gpu.load_fft(fft)
    startr = setup.rlo + 3735.0 * int((r - setup.rlo) // 3735.0)
    blk, rlo = gpu.subharm_ffdot_plane(startr)
    zi, col = (z + zmax) // 2, int(round(2 * (r - rlo)))
    frac = blk[zi, col] / (N * N / 4.0)
    assert 0.97 <= frac <= 1.002, frac
    assert blk.max() <= blk[zi, col] * (1 + 1e-6)
    hits, cands, sig, added = gpu.search()
    best = cands[np.argmax(cands["power"])]
    assert (best["r"], best["z"]) == (r, float(z))
