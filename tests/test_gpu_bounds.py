"""The harmonic sum over the plane's 16-bit upper bounds (k_harmsum_b / k_harmsum_bw / k_harmsum_bl,
and the coarse bound over the maxima of four columns in front of them: k_prefilter + k_flagged_exact,
hs_bound = 4) against the plain sum over the float plane (k_harmsum): they must return the same hits
bit for bit on every geometry, through every way the plane can be written, and at the values where
a rounded-up 16-bit bound behaves specially (equal to its power, overflowing to infinity).
The plain sum itself is checked against the CPU oracle in test_gpu_parity.py."""
import numpy as np
import pytest

import presto_amd as pa
from presto_amd import synth
from presto_amd.api import plan

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    g = pa.AccelSearchGPU(0)
    yield g
    g.set_tuning("hs_bound", 1)
    g.close()


def _hits_by_mode(gpu, modes=(0, 1, 2, 3, 4), max_hits=1 << 22, call=None, after_mode=None):
    """The search in every mode of the harmonic sum.  (Switching between 0 and the others changes
    the plane's layout, which sets the search up afresh: `after_mode` re-applies what that resets.)"""
    out = {}
    for m in modes:
        gpu.set_tuning("hs_bound", m)
        if after_mode:
            after_mode()
        out[m] = (call or gpu.run)(max_hits=max_hits)
    return out


def _configure_with_thresholds(gpu, numbins, N, dt, zmax, numharm, powcut, **kw):
    """A search whose thresholds are set by hand (so that a test decides how many cells cross)."""
    setup = plan(numbins, N, dt, zmax, numharm, **kw)
    for s in range(setup.numharmstages):
        setup.powcut[s] = powcut[s] if hasattr(powcut, "__len__") else powcut * (s + 1)
    gpu.create_subharminfos(setup.zmax)
    return gpu.set_search(setup)


def _assert_same(res):
    ref = res[0]
    for m, h in res.items():
        assert len(h) == len(ref) and h.tobytes() == ref.tobytes(), "hs_bound=%d differs from the float-plane sum" % m


@pytest.mark.parametrize("numharm", [1, 2, 4, 8, 16, 32])
@pytest.mark.parametrize("zmax", [0, 20, 200])
def test_same_hits_for_every_number_of_harmonics(gpu, numharm, zmax):
    N, dt = 1 << 20, 1e-4
    fft = synth.make_fft(N, dt, [dict(f=123.456, z=3.0, amp=0.3, duty=0.08), dict(f=741.0, z=-9.0, amp=0.2, duty=0.05)],
                         seed=300 + numharm)
    gpu.configure(N // 2, N, dt, zmax, numharm, sigma=1.5, fhi=4000.0)
    gpu.load_fft(fft)
    res = _hits_by_mode(gpu)
    assert len(res[0]) > 50           # a low threshold: many rows take the exact path
    _assert_same(res)


@pytest.mark.parametrize("numharm", [2, 4, 8, 16])
@pytest.mark.parametrize("zmax", [0, 50])
def test_coarse_bound_one_or_two_groups_per_lane(gpu, numharm, zmax):
    """The coarse bound's kernel with one group of four columns per lane and with two (pf_groups picks; left
    alone, two up to 8 harmonics and one with 16) must end in the same hits as the float-plane sum."""
    N, dt = 1 << 20, 1e-4
    fft = synth.make_fft(N, dt, [dict(f=223.456, z=5.0, amp=0.3, duty=0.08), dict(f=1741.0, z=-19.0, amp=0.2, duty=0.05)],
                         seed=400 + numharm)
    gpu.configure(N // 2, N, dt, zmax, numharm, sigma=1.5, flo=11.0, fhi=4900.0)
    gpu.load_fft(fft)
    try:
        res = _hits_by_mode(gpu, modes=(0,))
        for groups in (0, 1, 2):
            gpu.set_tuning("pf_groups", groups)
            res[10 + groups] = _hits_by_mode(gpu, modes=(4,))[4]
            assert gpu.get_tuning("hs_prefilter_active") == 1 and gpu.get_tuning("pf_listed") > 0
    finally:
        gpu.set_tuning("pf_groups", 0)
    assert len(res[0]) > 50
    for m, h in res.items():
        assert h.tobytes() == res[0].tobytes(), "pf_groups=%d differs from the float-plane sum" % (m - 10)


@pytest.mark.parametrize("numharm", [8, 32])
def test_ragged_end_and_odd_geometry(gpu, numharm):
    """A search whose column count is not a multiple of the kernels' tiles (dead lanes in the
    last wave, the later columns of a lane beyond the last searched one) and that starts high."""
    N, dt = 1 << 19, 1e-4
    fft = synth.noise_fft(N // 2, 77)
    _configure_with_thresholds(gpu, N // 2, N, dt, 10, numharm, [9.0, 12.0, 17.0, 26.0, 41.0, 68.0], flo=37.0, fhi=1913.0)
    gpu.load_fft(fft)
    res = _hits_by_mode(gpu)
    assert len(res[0]) > 300
    _assert_same(res)


@pytest.mark.parametrize("numharm", [16, 32])
def test_dense_hits(gpu, numharm):
    """A threshold so low that most rows of most waves cross it: the exact path is the common one."""
    N, dt = 1 << 18, 1e-4
    fft = synth.noise_fft(N // 2, 78)
    _configure_with_thresholds(gpu, N // 2, N, dt, 6, numharm, [2.5, 4.0, 7.0, 12.0, 21.0, 38.0], fhi=2000.0)
    gpu.load_fft(fft)
    res = _hits_by_mode(gpu, max_hits=1 << 24)
    assert len(res[0]) > 200000
    _assert_same(res)


@pytest.mark.parametrize("numharm", [16, 32])
def test_plane_written_by_the_host(gpu, numharm):
    """put_plane writes float cells into whichever layout the plane has -- planted powers far above
    every threshold, at values whose upper half alone is the power (1024.0), whose bound
    overflows (3.3e38 -> infinity, infinity -> NaN) or that sit one float below the next upper
    half (nextafter(2048, 0)) -- and values plane generation never writes but a host may:
    negative (the bounds hold for p >= 0 only: the kernel must notice) and NaN; neither is a hit,
    and neither may hide one."""
    N, dt = 1 << 19, 1e-4
    fft = synth.noise_fft(N // 2, 79)
    res, no_neg = {}, {}
    for mode in (0, 1, 2, 3, 4):
        gpu.set_tuning("hs_bound", mode)
        setup = gpu.configure(N // 2, N, dt, 20, numharm, sigma=6.0, fhi=2000.0)
        gpu.load_fft(fft)
        gpu.fill_plane()
        before = gpu.search_plane()
        col0 = 2 * setup.rlo
        plant = {(3, col0 + 7470 * 2 + 11): np.float32(1024.0),
                 (10, col0 + 7470 * 5 + 4001): np.float32(3.3e38),
                 (12, col0 + 7470 * 6 + 17): np.float32(np.inf),
                 (17, col0 + 7470 * 9 + 64): np.nextafter(np.float32(2048.0), np.float32(0.0))}
        for (zi, col), val in plant.items():
            gpu.put_plane(zi, col, np.array([val], dtype=np.float32))
            assert gpu.get_plane(zi, col, 1)[0] == val
        # a NaN alone (no negative power yet, so the sums over the bounds are still in use): not a
        # hit, and the hits of the rows it enters into are those of the float-plane sum
        gpu.put_plane(7, col0 + 7470 * 4 + 333, np.array([np.nan], dtype=np.float32))
        no_neg[mode] = gpu.search_plane()
        neg = np.array([-1.0000001, np.nan, -3.5e-5], dtype=np.float32)
        gpu.put_plane(5, col0 + 7470 * 3 + 100, neg)
        assert gpu.get_plane(5, col0 + 7470 * 3 + 100, 3).tobytes() == neg.tobytes()
        res[mode] = gpu.search_plane()
        got = {(int(h["zind"]), int(h["rind"])): h["power"] for h in res[mode] if h["stage"] == 0}
        for key, val in plant.items():
            assert key in got and got[key] == val
        assert len(res[mode]) > len(before)
    _assert_same(res)
    _assert_same(no_neg)
    assert len(no_neg[3]) >= len(res[3])


def test_coarse_bound_list_overflow_and_long_transforms(gpu):
    """The coarse bound's list of (row, group)s to sum exactly has a capacity: beyond it the plain
    bounded sum runs instead (same hits).  And with 16384-point transforms (zmax 1200, 400) plane
    generation writes the group maxima from single stores (four neighbouring lanes hold a group)."""
    N, dt = 1 << 19, 1e-4
    fft = synth.noise_fft(N // 2, 84)
    _configure_with_thresholds(gpu, N // 2, N, dt, 10, 16, [9.0, 12.0, 17.0, 26.0, 41.0], fhi=2000.0)
    gpu.load_fft(fft)
    try:
        gpu.set_tuning("pf_list_cap", 10)
        res = _hits_by_mode(gpu, modes=(0, 4))
        assert gpu.get_tuning("pf_listed") == -1          # the list was of no use
        _assert_same(res)
    finally:
        gpu.set_tuning("pf_list_cap", 0)
    res = _hits_by_mode(gpu, modes=(0, 4))
    assert len(res[0]) > 100 and gpu.get_tuning("pf_listed") > 10
    _assert_same(res)
    # a threshold so low that a workgroup's own list overflows into the global one
    _configure_with_thresholds(gpu, N // 2, N, dt, 10, 16, [5.0, 7.0, 10.0, 16.0, 27.0], fhi=2000.0)
    gpu.load_fft(fft)
    res = _hits_by_mode(gpu, modes=(0, 4), max_hits=1 << 24)
    assert len(res[0]) > 20000 and gpu.get_tuning("pf_listed") > 4096
    _assert_same(res)
    N = 1 << 20
    fft = synth.make_fft(N, dt, [dict(f=300.3, z=-40.0, amp=0.3, duty=0.1)], seed=85)
    setup = gpu.configure(N // 2, N, dt, 1200, 8, sigma=2.0, fhi=4000.0)
    assert setup.fftlen == 16384
    gpu.load_fft(fft)
    res = _hits_by_mode(gpu, modes=(0, 1, 4))
    assert len(res[0]) > 50 and gpu.get_tuning("pf_listed") >= 0
    _assert_same(res)
    for numharm, lo in ((16, 37.0), (4, 1.0)):   # another shift (zmax 400: 480 outputs dropped in front), a search that starts high
        setup = gpu.configure(N // 2, N, dt, 400, numharm, sigma=1.5, flo=lo, fhi=4000.0)
        assert setup.fftlen == 16384
        gpu.load_fft(fft)
        res = _hits_by_mode(gpu, modes=(0, 3, 4))
        assert len(res[0]) > 50 and gpu.get_tuning("hs_prefilter_active") == 1 and gpu.get_tuning("pf_listed") > 0
        _assert_same(res)


def test_block_range_and_chunked_sum(gpu):
    N, dt = 1 << 21, 1e-4
    fft = synth.make_fft(N, dt, [dict(f=200.2, z=5.0, amp=0.3, duty=0.1)], seed=81)
    setup = gpu.configure(N // 2, N, dt, 20, 8, sigma=2.0, fhi=4900.0)
    gpu.load_fft(fft)
    whole = _hits_by_mode(gpu)
    _assert_same(whole)
    nb = setup.num_search_blocks
    try:
        part = _hits_by_mode(gpu, after_mode=lambda: gpu.set_block_range(nb // 3, nb // 2))
        _assert_same(part)
        want = whole[0][(whole[0]["block"] >= nb // 3) & (whole[0]["block"] < nb // 3 + nb // 2)]
        assert part[0].tobytes() == want.tobytes()
    finally:
        gpu.set_block_range(0, nb)
    try:
        gpu.set_tuning("hs_chunk_blocks", 40)
        chunked = _hits_by_mode(gpu)
        _assert_same(chunked)
        assert chunked[1].tobytes() == whole[0].tobytes()
    finally:
        gpu.set_tuning("hs_chunk_blocks", 0)


def test_changing_the_layout_between_trials(gpu):
    """hs_bound 0 <-> 1 changes the plane's layout (floats <-> two planes of halves): the plane is
    empty afterwards, the next trial fills it again, and nothing of the other layout's trial is
    left; the table-form plane generation (corr_rr = 0) exists for the float layout only and the
    knob is ignored in the other."""
    N, dt = 1 << 20, 1e-4
    a = synth.make_fft(N, dt, [dict(f=150.0, z=2.0, amp=0.3, duty=0.1)], seed=82)
    b = synth.make_fft(N, dt, [dict(f=333.3, z=-6.0, amp=0.3, duty=0.1)], seed=83)
    gpu.set_tuning("hs_bound", 0)
    gpu.configure(N // 2, N, dt, 20, 16, sigma=2.0, fhi=4000.0)
    gpu.load_fft(a)
    want_a = gpu.run()
    gpu.load_fft(b)
    want_b = gpu.run()
    assert want_a.tobytes() != want_b.tobytes()
    gpu.set_tuning("hs_bound", 1)
    assert len(gpu.search_plane()) == 0          # the new layout's plane is empty
    gpu.load_fft(a)
    assert gpu.run().tobytes() == want_a.tobytes()
    gpu.set_tuning("hs_bound", 0)
    assert len(gpu.search_plane()) == 0
    gpu.load_fft(b)
    assert gpu.run().tobytes() == want_b.tobytes()
    gpu.set_tuning("hs_bound", 1)
    gpu.load_fft(b)
    assert gpu.run().tobytes() == want_b.tobytes()
    try:
        gpu.set_tuning("corr_rr", 0)
        gpu.load_fft(a)
        assert gpu.run().tobytes() == want_a.tobytes()
    finally:
        gpu.set_tuning("corr_rr", 1)


def test_knob_reports_whether_the_bounds_are_in_use(gpu):
    N, dt = 1 << 19, 1e-4
    gpu.set_tuning("hs_bound", 1)
    gpu.configure(N // 2, N, dt, 10, 8, fhi=1000.0)
    assert gpu.get_tuning("hs_bound_active") == 1
    gpu.set_tuning("hs_bound", 0)
    assert gpu.get_tuning("hs_bound_active") == 0
    gpu.configure(N // 2, N, dt, 10, 8, fhi=1000.0, inmem=False)      # standard mode: no plane, no bounds
    gpu.set_tuning("hs_bound", 1)
    assert gpu.get_tuning("hs_bound_active") == 0
    gpu.configure(N // 2, N, dt, 10, 8, fhi=1000.0)
    assert gpu.get_tuning("hs_bound_active") == 1


def test_negative_zero_in_the_plane_counts_as_negative(gpu):
    """-0.0 written by a host: its upper half (0x8000) is larger, as a 16-bit number, than that of any positive
    power, so the maximum of its group of four would hide a crossing next to it -- unless the cell raises the
    'negative power' flag like any other negative value, and every row takes the exact path."""
    N, dt = 1 << 19, 1e-4
    fft = synth.noise_fft(N // 2, 91)
    setup = _configure_with_thresholds(gpu, N // 2, N, dt, 10, 8, [30.0, 40.0, 50.0, 60.0], fhi=2000.0)
    gpu.load_fft(fft)
    col = 2 * setup.rlo + 7470 * 3 + 1001
    res = {}
    try:
        for m in (0, 4, 1):
            gpu.set_tuning("hs_bound", m)
            gpu.fill_plane()
            row = gpu.get_plane(4, col - 2, 8)
            row[2] = np.float32(-0.0)
            row[3] = np.float32(1000.0)      # a crossing in the same group of four as the -0.0 cell, or the next one
            row[5] = np.float32(900.0)
            gpu.put_plane(4, col - 2, row)
            res[m] = gpu.search_plane(max_hits=1 << 20)
    finally:
        gpu.set_tuning("hs_bound", 1)
    assert len(res[0]) >= 2 and np.any(res[0]["power"] >= 1000.0)
    _assert_same(res)
    # the next whole plane generation rewrites every cell a host may have touched: the bounds hold again
    gpu.set_tuning("hs_bound", 4)
    try:
        a = gpu.run(max_hits=1 << 20)
        assert gpu.get_tuning("pf_listed") >= 0
        gpu.set_tuning("hs_bound", 0)
        b = gpu.run(max_hits=1 << 20)
    finally:
        gpu.set_tuning("hs_bound", 1)
    assert a.tobytes() == b.tobytes()


def test_coarse_bound_is_left_out_after_it_was_given_up(gpu):
    """Thresholds so low that the coarse bound's list would be long: a sample taken before the walk says so, the
    bounded sum runs over the whole plane, and the context goes without the coarse bound for its next launches
    (4, then 8 ...), trying again after them.  Same hits throughout."""
    N, dt = 1 << 23, 64e-6
    fft = synth.noise_fft(N // 2, 92)
    setup = plan(N // 2, N, dt, 20, 8, fhi=1e9)
    for s in range(setup.numharmstages):
        setup.powcut[s] *= 0.6
    gpu.create_subharminfos(setup.zmax)
    gpu.set_search(setup)
    gpu.load_fft(fft)
    gpu.set_tuning("hits_cap", 1 << 22)
    gpu.set_tuning("hs_bound", 3)
    ref = gpu.run(max_hits=1 << 24)
    gpu.set_tuning("hs_bound", 1)
    try:
        seen = []
        for k in range(7):
            assert gpu.get_tuning("hs_prefilter_active") == 1 or gpu.get_tuning("pf_skip") >= 0
            h = gpu.run(max_hits=1 << 24)
            assert h.tobytes() == ref.tobytes()
            seen.append((gpu.get_tuning("pf_last_listed"), gpu.get_tuning("pf_skip")))
        assert seen[0] == (-1, 4), seen                     # given up; the next four go without
        assert [s[0] for s in seen[1:5]] == [-2] * 4, seen  # ... the coarse bound did not run
        assert seen[5] == (-1, 8), seen                     # tried again, given up again: eight
        gpu.set_tuning("pf_memory", 0)                      # every launch tries
        h = gpu.run(max_hits=1 << 24)
        assert h.tobytes() == ref.tobytes() and gpu.get_tuning("pf_last_listed") == -1 and gpu.get_tuning("pf_skip") == 0
        # without the sample the walk runs to the end of its list's room and gives up there: same hits
        gpu.set_tuning("pf_sample", 0)
        h = gpu.run(max_hits=1 << 24)
        assert h.tobytes() == ref.tobytes() and gpu.get_tuning("pf_last_listed") == -1
    finally:
        gpu.set_tuning("pf_memory", 1)
        gpu.set_tuning("pf_sample", 1)
        gpu.set_tuning("hits_cap", 1 << 20)


def test_group_maxima_of_more_blocks_than_a_grid_dimension_counts(gpu):
    """A series so long that it has more blocks than the 65 535 a launch grid's second dimension can count:
    the kernel that takes the group maxima from the plane (a host has written into it) folds the blocks into
    the grid's first dimension.  zmax 0, so that the plane stays small."""
    numbins = 65600 * 3735 + 4000
    N, dt = 2 * numbins, 64e-6
    rng = np.random.default_rng(93)
    fft = np.empty(numbins, np.complex64)
    step = 1 << 24
    for a in range(0, numbins, step):                       # unit-variance complex noise, written in pieces
        n = min(step, numbins - a)
        fft[a:a + n] = (rng.standard_normal(n, np.float32) + 1j * rng.standard_normal(n, np.float32)) * np.float32(0.7071)
    setup = plan(numbins, N, dt, 0, 8, sigma=2.0, flo=1.0, fhi=1e9)
    for s in range(setup.numharmstages):
        setup.powcut[s] *= 0.75                               # a handful of noise cells cross
    gpu.create_subharminfos(setup.zmax)
    setup = gpu.set_search(setup)
    assert setup.num_search_blocks + setup.num_prepop_blocks > 65535
    gpu.load_fft(fft)
    try:
        gpu.set_tuning("hs_bound", 3)
        ref = gpu.run(max_hits=1 << 22)
        gpu.set_tuning("hs_bound", 4)
        a = gpu.run(max_hits=1 << 22)                       # maxima from plane generation
        cell = gpu.get_plane(0, 2 * setup.rlo + 12345, 1)
        gpu.put_plane(0, 2 * setup.rlo + 12345, cell)       # the same value back: the maxima are taken from the plane now
        b = gpu.search_plane(max_hits=1 << 22)
        assert gpu.get_tuning("pf_listed") >= 0
    finally:
        gpu.set_tuning("hs_bound", 1)
    assert len(ref) > 0 and a.tobytes() == ref.tobytes() and b.tobytes() == ref.tobytes()
