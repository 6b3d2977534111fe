"""CPU-only tests: the C ABI loads and exports what include/accelgpu.h declares,
the host-side logic of the library (planning, detection statistics, candidate
replay) agrees with the oracle, and the oracle reproduces its golden vectors."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "search_small.npz")


def test_cabi_exports_every_declared_symbol():
    import presto_amd as pa
    hdr = open(os.path.join(ROOT, "include", "accelgpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(accelgpu_[a-z_0-9]+)\s*\(", hdr)))
    assert len(names) >= 25
    lib = C.CDLL(pa.api.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert pa.load_library().accelgpu_version().decode().startswith("accelgpu")


def test_struct_layouts_match_header():
    import presto_amd as pa
    assert C.sizeof(pa.api.Setup) == 8 * 4 + 4 * 8 + 8 * 6 + 4 + 4 * 6 + 4 + 8 * 6 + 16
    assert pa.HIT_DTYPE.itemsize == 24 and pa.CAND_DTYPE.itemsize == 32


@pytest.mark.parametrize("numbins,dt,zmax,numharm,flo,fhi,lobin", [
    (524288, 1e-4, 0, 8, 1.0, 1e9, 0), (4194304, 64e-6, 200, 16, 1.0, 1e9, 0),
    (67108864, 64e-6, 300, 16, 1.0, 1e9, 0), (8388608, 64e-6, 1200, 32, 1.0, 1e9, 0),
    (100000, 1e-3, 51, 4, 0.5, 40.0, 0), (100000, 1e-3, 10, 1, 0.0, 1e9, 300)])
def test_plan_matches_oracle(oracle, numbins, dt, zmax, numharm, flo, fhi, lobin):
    import presto_amd as pa
    p = pa.plan(numbins, 2 * numbins, dt, zmax, numharm, 2.0, flo, fhi, lobin)
    o = oracle.Obs()
    dummy = np.zeros(4, dtype=np.complex64)
    assert oracle.lib().orc_obs_init(C.byref(o), dummy.ctypes.data_as(C.c_void_p), numbins,
                                     2 * numbins, dt, zmax, numharm, 2.0, flo, fhi, lobin, 0, 0, 0) == 0
    assert (p.rlo, p.rhi, p.highestbin, p.numz, p.fftlen, p.numharmstages, p.rlen, p.zmax) == \
        (o.rlo, o.rhi, o.highestbin, o.numz, o.fftlen, o.numharmstages, o.rlen, o.zmax)
    assert list(p.numindep)[:p.numharmstages] == list(o.numindep)[:o.numharmstages]
    assert list(p.powcut)[:p.numharmstages] == list(o.powcut)[:o.numharmstages]
    assert p.num_search_blocks == oracle.lib().orc_num_search_blocks(C.byref(o))
    assert p.num_prepop_blocks == oracle.lib().orc_num_prepop_blocks(C.byref(o))


def test_plan_rejects_what_the_reference_rejects():
    import presto_amd as pa
    for kw in (dict(numharm=3), dict(numharm=64), dict(flo=1e9), dict(lobin=10 ** 7)):
        args = dict(numbins=1 << 16, N=1 << 17, dt=1e-3, zmax=20, numharm=8)
        args.update(kw)
        with pytest.raises(pa.AccelGPUError):
            pa.plan(**args)
    with pytest.raises(pa.AccelGPUError):                       # fhi below flo
        pa.plan(1 << 16, 1 << 17, 1e-3, 20, 8, flo=50.0, fhi=10.0)
    assert pa.plan(1 << 16, 1 << 17, 1e-3, 21, 8).zmax == 22    # odd zmax rounds up


def test_sigma_functions_match_oracle(oracle):
    import presto_amd.api as api
    L = oracle.lib()
    rng = np.random.default_rng(0)
    for nh in (1, 2, 4, 8, 16, 32):
        for nt in (1e3, 15236347.0, 5831448272.0):
            assert api.power_for_sigma(2.0, nh, nt) == L.orc_power_for_sigma(2.0, nh, nt)
            for pw in np.concatenate([rng.uniform(0.1 * nh, 40 * nh, 50), [0.0, -1.0, 15.0 * nh, 15.0001 * nh]]):
                # both solve the same equation by Newton steps, from different starting points:
                # equal to a few units in the last place of a double (the search uses float sigmas)
                got, want = api.candidate_sigma(pw, nh, nt), L.orc_candidate_sigma(float(pw), nh, nt)
                assert abs(got - want) <= 4e-15 * max(1.0, abs(want)), (pw, nh, nt, got, want)


def test_oracle_reproduces_golden(oracle):
    g = np.load(GOLD)
    s = oracle.Search(g["fft"], int(g["N"]), float(g["dt"]), int(g["zmax"]), int(g["numharm"]), fhi=1e9)
    assert np.array_equal(s.powcut, g["powcut"]) and np.array_equal(s.numindep, g["numindep"])
    hits, cands = s.run()
    assert len(hits) == len(g["hits"]) and len(cands) == len(g["cands"])
    for f in ("numharm", "block", "zind", "rind", "added", "r", "z"):
        assert np.array_equal(hits[f], g["hits"][f]), f
    np.testing.assert_allclose(hits["power"], g["hits"]["power"], rtol=2e-6)
    np.testing.assert_allclose(hits["sigma"], g["hits"]["sigma"], rtol=2e-5, atol=1e-5)
    assert np.array_equal(cands["r"], g["cands"]["r"]) and np.array_equal(cands["z"], g["cands"]["z"])
    blk, _ = s.fund_block(s.rlo)
    np.testing.assert_allclose(blk[::5, ::37], g["block0_rows"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(oracle.init_kernel(10, s.fftlen)[0][::16], g["kernel_z10"], rtol=0, atol=1e-6)
    s.close()


def test_replay_matches_oracle_list(oracle):
    """The library's array-based candidate replay against the oracle's linked
    list, on the golden hit sequence and on adversarial synthetic sequences."""
    import presto_amd as pa
    from presto_amd.api import replay_candidates
    g = np.load(GOLD)
    setup = pa.plan(g["fft"].size, int(g["N"]), float(g["dt"]), int(g["zmax"]), int(g["numharm"]), fhi=1e9)
    gh = g["hits"]
    hits = np.zeros(len(gh), dtype=pa.HIT_DTYPE)
    hits["power"], hits["zind"], hits["rind"], hits["block"] = gh["power"], gh["zind"], gh["rind"], gh["block"]
    hits["stage"] = np.log2(gh["numharm"]).astype(np.int32)
    cands, sig, added = replay_candidates(setup, hits)
    assert np.array_equal(added, gh["added"])
    assert np.array_equal(sig, gh["sigma"])
    gc = g["cands"]
    assert len(cands) == len(gc)
    for f in ("r", "z", "numharm", "power", "sigma"):
        assert np.array_equal(cands[f], gc[f]), f


def test_replay_random_sequences_against_python_model():
    """Independent pure-Python model of the list discipline (small cases)."""
    import presto_amd as pa
    from presto_amd.api import replay_candidates
    setup = pa.plan(1 << 16, 1 << 17, 1e-3, 20, 8, fhi=1e9)
    rng = np.random.default_rng(9)
    for trial in range(20):
        n = 300
        hits = np.zeros(n, dtype=pa.HIT_DTYPE)
        hits["stage"] = rng.integers(0, 4, n)
        hits["zind"] = rng.integers(0, 21, n)
        hits["rind"] = rng.integers(200, 3000, n)          # dense: lots of 15-bin collisions
        hits["power"] = rng.uniform(20, 80, n).astype(np.float32)
        cands, sig, added = replay_candidates(setup, hits)
        lst = []                                            # [r, sigma, idx]
        want_added = []
        for i in range(n):
            nh = 1 << int(hits["stage"][i])
            rr = hits["rind"][i] * 0.5 / nh
            s = sig[i]
            a = 0
            if not lst:
                lst.append([rr, s, i]); a = 1
            else:
                cur = 0
                while cur < len(lst) - 1 and lst[cur][0] < rr:
                    cur += 1
                nd = abs(rr - lst[cur][0])
                pd = abs(rr - lst[cur - 1][0]) if cur > 0 else 16.0
                if pd < 15.0:
                    if lst[cur - 1][1] < s:
                        lst[cur - 1] = [rr, s, i]; a = 1
                    if nd < 15.0 and lst[cur][1] < s:
                        if a:
                            del lst[cur]
                        else:
                            lst[cur] = [rr, s, i]; a = 1
                elif nd < 15.0:
                    if lst[cur][1] < s:
                        lst[cur] = [rr, s, i]; a = 1
                else:
                    a = 1
                    if cur == len(lst) - 1 and lst[cur][0] < rr - 15.0:
                        lst.append([rr, s, i])
                    else:
                        lst.insert(cur, [rr, s, i])
            want_added.append(a)
        assert np.array_equal(added, np.array(want_added))
        assert [c for c in cands["added_order"]] == [e[2] for e in lst]


def test_synthetic_inf_text_round_trip(tmp_path):
    from presto_amd import synth
    fft = synth.noise_fft(1024, 1)
    synth.write_fft(str(tmp_path / "x"), fft, 2048, 1e-3)
    assert os.path.getsize(tmp_path / "x.fft") == 1024 * 8
    txt = open(tmp_path / "x.inf").read()
    assert "Number of bins in the time series" in txt and "=  2048" in txt


def test_read_birds_follows_the_reference_rules(tmp_path):
    """get_birdies (reference src/birdzap.c:6-88): comments, barycentric 'B' lines, Doppler
    factor on the others, minimum width of one bin, sorted by low edge."""
    import presto_amd.api as api
    p = tmp_path / "birds.zaplist"
    p.write_text("# freq width\n60.0 0.1\nB 10.0 0.02\n120.0 0.0001\n# done\n 0.5 0.01\n")
    T, v = 1000.0, 1e-4
    lo, hi = api.read_birds(str(p), T, v)
    assert len(lo) == 4 and np.all(np.diff(lo) > 0)
    k = 1.0 + v
    want = sorted([((60.0 - 0.05) * T * k, (60.0 + 0.05) * T * k), ((10.0 - 0.01) * T, (10.0 + 0.01) * T),
                   (120.0 * T * k - 0.5, 120.0 * T * k + 0.5), ((0.5 - 0.005) * T * k, (0.5 + 0.005) * T * k)])
    assert np.allclose(lo, [w[0] for w in want], rtol=0, atol=1e-7)
    assert np.allclose(hi, [w[1] for w in want], rtol=0, atol=1e-7)


POST = os.path.join(ROOT, "tests", "golden", "post_small.npz")


def test_oracle_reproduces_post_golden(oracle, tmp_path):
    """The committed vectors of the tail and the pre-steps (tests/golden/post_small.npz and
    the ACCEL text beside it) still come out of the oracle bit for bit."""
    g = np.load(GOLD)
    p = np.load(POST)
    s = oracle.Search(g["fft"], int(g["N"]), float(g["dt"]), int(g["zmax"]), int(g["numharm"]), fhi=1e9)
    hits, cands = s.run()
    opt = s.optimize(cands, polish=True, eliminate=True)
    assert opt.tobytes() == p["opt"].tobytes()
    name = str(tmp_path / "post_small_DM7.50_ACCEL_30")
    from presto_amd import synth
    s.write_accel_files(opt, name, name + ".cand", synth.inf_text("post_small_DM7.50", int(g["N"]), float(g["dt"])))
    gold = os.path.join(ROOT, "tests", "golden", "post_small_DM7.50_ACCEL_30")
    assert open(name).read() == open(gold).read()
    assert open(name + ".cand", "rb").read() == open(gold + ".cand", "rb").read()
    spec = oracle.realfft(p["series"])
    assert np.array_equal(spec, p["realfft"])
    assert np.array_equal(oracle.deredden(spec), p["deredden"])
    assert np.array_equal(oracle.rednoise(spec, 8192 * 1e-3), p["rednoise"])
    assert np.array_equal(oracle.zapbirds(spec, p["birds_lo"], p["birds_hi"]), p["zapped"])


def test_product_writer_reproduces_golden_accel_file(tmp_path):
    """The library's writer on the golden optimised candidates gives the golden ACCEL text
    and .cand bytes; the sifting parser reads it back."""
    import presto_amd as pa
    import presto_amd.api as api
    from presto_amd import sifting, synth
    g = np.load(GOLD)
    p = np.load(POST)
    N, dt = int(g["N"]), float(g["dt"])
    setup = pa.plan(N // 2, N, dt, int(g["zmax"]), int(g["numharm"]), fhi=1e9)
    opt = np.zeros(len(p["opt"]), dtype=api.OPTCAND_DTYPE)
    for k in opt.dtype.names:
        opt[k] = p["opt"][k]
    name = str(tmp_path / "post_small_DM7.50_ACCEL_30")
    api.write_accel_files(setup, opt, name, name + ".cand", synth.inf_text("post_small_DM7.50", N, dt))
    gold = os.path.join(ROOT, "tests", "golden", "post_small_DM7.50_ACCEL_30")
    assert open(name).read() == open(gold).read()
    assert open(name + ".cand", "rb").read() == open(gold + ".cand", "rb").read()
    cl = sifting.candlist_from_candfile(gold)
    assert len(cl) == len(opt) and cl[0].DM == 7.5 and cl[0].numharm == opt["numharm"][0]


def test_replay_binary_search_equals_the_plain_walk():
    """The candidate-list replay finds its place in the list by a binary search over
    running maxima; the reference walks from the head.  Same lists, same per-hit flags,
    on clustered hits that overwrite, erase and insert out of order."""
    import presto_amd as pa
    from presto_amd.api import replay_candidates, replay_options, plan
    setup = plan(1 << 22, 1 << 23, 64e-6, 200, 16, fhi=1e9)
    rng = np.random.default_rng(5)
    for trial, (n, ncen, spread) in enumerate([(20000, 60, 40), (30000, 700, 12), (5000, 5, 300), (3000, 3000, 1)]):
        hits = np.zeros(n, dtype=pa.HIT_DTYPE)
        hits["stage"] = rng.integers(0, 5, n)
        hits["power"] = (np.array(setup.powcut)[hits["stage"]] * rng.uniform(1.0, 2.5, n)).astype(np.float32)
        hits["zind"] = rng.integers(0, 201, n)
        centres = rng.uniform(1e3, 8e6, ncen)
        hits["rind"] = (centres[rng.integers(0, ncen, n)] + rng.integers(-spread, spread + 1, n)).astype(np.int64)
        hits["block"] = hits["rind"] // 7470
        hits = hits[np.lexsort((hits["rind"], hits["zind"], hits["stage"], hits["block"]))]
        replay_options(linear_walk=True, threads=1)
        c0, s0, a0 = replay_candidates(setup, hits)
        replay_options()
        c1, s1, a1 = replay_candidates(setup, hits)
        assert len(c0) > 2 and np.array_equal(c0, c1), trial
        assert np.array_equal(s0, s1) and np.array_equal(a0, a1)
        # and in an order the search never produces (r not even nearly sorted)
        perm = rng.permutation(n)
        replay_options(linear_walk=True, threads=1)
        c0, s0, a0 = replay_candidates(setup, hits[perm])
        replay_options()
        c1, s1, a1 = replay_candidates(setup, hits[perm])
        assert np.array_equal(c0, c1) and np.array_equal(a0, a1)


def test_coarse_bound_shape_predicate():
    """Where the coarse bound in front of the harmonic sum may run: its kernels count columns in 32 bits, so a
    plane row of 2^32 columns or more must be refused whatever else holds (forced included); 2 to 16 harmonics
    on at least 2^22 search columns of 8192-point transforms otherwise."""
    import presto_amd as pa
    ok = pa.load_library().accelgpu_coarse_bound_applies
    cols, rlen = 1 << 23, (1 << 23) + 20000
    assert ok(5, cols, rlen, 8192, 0) == 1 and ok(2, cols, rlen, 8192, 0) == 1
    assert ok(1, cols, rlen, 8192, 0) == 0 and ok(1, cols, rlen, 8192, 1) == 0      # nothing to sum
    assert ok(6, cols, rlen, 8192, 0) == 0 and ok(6, cols, rlen, 8192, 1) == 1      # 32 harmonics: only when forced
    assert ok(5, (1 << 22) - 1, rlen, 8192, 0) == 0 and ok(5, 1 << 22, rlen, 8192, 0) == 1
    assert ok(5, cols, rlen, 16384, 0) == 1 and ok(5, cols, rlen, 16384, 1) == 1    # both transform lengths write the maxima
    assert ok(4, cols, rlen, 16384, 0) == 1 and ok(3, cols, rlen, 16384, 0) == 0    # (the long one pays with 8 and 16 harmonics)
    assert ok(3, cols, rlen, 16384, 1) == 1
    assert ok(5, cols, rlen, 4096, 0) == 0 and ok(5, cols, rlen, 4096, 1) == 1      # others: maxima from the plane when forced
    for forced in (0, 1):
        assert ok(5, 1 << 33, (1 << 32) - 1, 8192, forced) == 1
        assert ok(5, 1 << 33, 1 << 32, 8192, forced) == 0
        assert ok(5, 1 << 33, 1 << 40, 8192, forced) == 0
