"""Cross-DM sifting (presto_amd/sifting.py) on ACCEL files written by the library.

The reference module (lib/python/sifting.py) is Python 2 and imports the compiled
presto extension, so it cannot run here: parity unpinned.  These are known-answer
tests of the rules it states: duplicates within r_err bins collapse onto the most
significant one and collect its hits, harmonics of a stronger candidate go, DM
problems (too few hits, peak at low DM, gaps in DM) go, thresholds reject.
"""
import numpy as np
import pytest

from presto_amd import sifting, synth


def _write_trial(tmp_path, dm, rows, N=1 << 22, dt=1e-4):
    """rows: (r, z, numharm, sigma, power per harmonic)."""
    import presto_amd as pa
    import presto_amd.api as api
    setup = pa.plan(N // 2, N, dt, 50, 8, fhi=1e9)
    c = np.zeros(len(rows), dtype=api.OPTCAND_DTYPE)
    rng = np.random.default_rng(int(dm * 100))
    for i, (r, z, nh, sigma, hp) in enumerate(rows):
        c["r"][i], c["z"][i], c["numharm"][i], c["sigma"][i] = r, z, nh, sigma
        c["power"][i] = hp * nh
        c["hirs"][i, :nh] = r * np.arange(1, nh + 1)
        c["hizs"][i, :nh] = z * np.arange(1, nh + 1)
        d = c["derivs"][i]
        d["locpow"][:nh] = 1.0
        d["pow"][:nh] = hp
        d["phs"][:nh] = rng.uniform(0.1, 6.0, nh)
        d["d2pow"][:nh] = -hp * 3.0
    name = str(tmp_path / ("beam_DM%.2f_ACCEL_50" % dm))
    api.write_accel_files(setup, c, name, name + ".cand", synth.inf_text("beam_DM%.2f" % dm, N, dt))
    return name, c, setup


def test_parse_power():
    assert sifting.parse_power("12.3(4)") == 12.3
    assert sifting.parse_power("1.2(3)x10^5") == pytest.approx(1.2e5)


def test_file_round_trip_and_in_memory_route(tmp_path):
    rows = [(12345.25, 4.0, 4, 12.0, 30.0), (555.5, 0.0, 1, 7.5, 60.0), (80000.125, -10.5, 8, 25.0, 9.0)]
    name, c, setup = _write_trial(tmp_path, 12.5, rows)
    cl = sifting.candlist_from_candfile(name)
    assert len(cl) == 3
    T = setup.T
    for cand, (r, z, nh, sigma, hp) in zip(cl, rows):
        assert cand.DM == 12.5 and cand.numharm == nh
        assert abs(cand.r - r) < 0.13 and abs(cand.z - z) <= 0.5 and cand.T == pytest.approx(T)
        assert cand.harm_pows == pytest.approx([hp] * nh, rel=0.02)
        assert cand.snr == pytest.approx(nh * np.sqrt(hp - 1.0), rel=0.02)
        assert cand.hits == [(12.5, cand.snr, cand.sigma)]
        assert cand.sigma == pytest.approx(sifting.candidate_sigma(sum(cand.harm_pows), nh, 1))
    mem = sifting.candlist_from_optcands(c, setup, "12.50", name)
    for a, b in zip(cl, mem):
        assert (a.candnum, a.numharm, a.r, a.z, a.ipow_det, a.cpow) == (b.candnum, b.numharm, b.r, b.z, b.ipow_det, b.cpow)
        assert a.sigma == b.sigma and a.snr == b.snr and np.array_equal(a.harm_pows, b.harm_pows)
        assert str(a) == str(b)


@pytest.mark.parametrize("nph", [0.0, 3.5e4])
def test_in_memory_route_equals_file_route_on_random_candidates(tmp_path, nph):
    """Every number the consumer reads, over five decades of power (plain and x10^n
    notation), negative z, up to 32 harmonics."""
    import presto_amd as pa
    import presto_amd.api as api
    N, dt = 1 << 23, 64e-6
    setup = pa.plan(N // 2, N, dt, 200, 32, fhi=1e9)
    rng = np.random.default_rng(11)
    n = 60
    c = np.zeros(n, dtype=api.OPTCAND_DTYPE)
    c["numharm"] = 2 ** rng.integers(0, 6, n)
    c["r"] = np.sort(rng.uniform(300, 3e6, n)) / c["numharm"]
    c["r"] = np.round(c["r"] * 2 * c["numharm"]) / (2 * c["numharm"])
    c["z"] = rng.integers(-100, 100, n) * 2.0 / c["numharm"]
    c["power"] = rng.uniform(20, 4000, n)
    c["sigma"] = rng.uniform(2, 60, n)
    for i in range(n):
        nh = int(c["numharm"][i])
        c["hirs"][i, :nh] = (c["r"][i] + rng.uniform(-0.2, 0.2)) * np.arange(1, nh + 1)
        c["hizs"][i, :nh] = (c["z"][i] + rng.uniform(-0.5, 0.5)) * np.arange(1, nh + 1)
        d = c["derivs"][i]
        d["locpow"][:nh] = rng.uniform(0.5, 2.0, nh) * (nph if nph else 1.0)
        d["pow"][:nh] = d["locpow"][:nh] * 10.0 ** rng.uniform(-0.5, 5.5, nh)
        d["phs"][:nh] = rng.uniform(0, 2 * np.pi, nh)
        d["dphs"][:nh] = rng.uniform(-3, 3, nh)
        d["d2pow"][:nh] = -d["pow"][:nh] * rng.uniform(0.5, 8, nh)
    name = str(tmp_path / "rand_DM33.00_ACCEL_200")
    api.write_accel_files(setup, c, name, name + ".cand", synth.inf_text("rand_DM33.00", N, dt), nph=nph)
    assert "x10^" in open(name).read()
    cl = sifting.candlist_from_candfile(name)
    mem = sifting.candlist_from_optcands(c, setup, "33.00", name, nph=nph)
    assert len(cl) == len(mem) == n
    for a, b in zip(cl, mem):
        assert (a.candnum, a.numharm, a.r, a.z, a.ipow_det, a.cpow, a.sigma, a.snr, a.DM, a.T) == \
               (b.candnum, b.numharm, b.r, b.z, b.ipow_det, b.cpow, b.sigma, b.snr, b.DM, b.T)
        assert np.array_equal(a.harm_pows, b.harm_pows) and np.array_equal(a.harm_amps, b.harm_amps)
        assert a.hits == b.hits and str(a) == str(b)


def test_sift_across_dm_trials(tmp_path):
    sifting.r_err, sifting.sigma_threshold = 1.1, 6.0
    dms = [0.0, 2.5, 5.0, 7.5, 10.0]
    P = 20000.3            # a pulsar: strongest at DM 5, seen at 2.5..7.5
    files = []
    for dm in dms:
        rows = []
        if dm in (2.5, 5.0, 7.5):
            rows.append((P + 0.2 * (dm - 5.0) / 2.5, 2.0, 4, 20.0 - abs(dm - 5.0), 40.0 - 4 * abs(dm - 5.0)))
            rows.append((2 * P + 0.1, 4.0, 2, 9.0, 25.0))                # its 2nd harmonic, weaker
        rows.append((60000.0, 0.0, 2, 15.0 - dm, 50.0 - 2 * dm))         # RFI: peaks at DM 0
        if dm == 10.0:
            rows.append((33333.3, 0.0, 2, 9.0, 30.0))                    # a one-DM wonder
        if dm in (2.5, 10.0):
            rows.append((45000.7, 0.0, 2, 10.0, 30.0 + dm))              # hits only at non-adjacent DMs
        rows.append((71000.0 + dm, 0.0, 1, 3.0, 8.5))                    # below both thresholds
        rows.append((150.0, 0.0, 2, 30.0, 80.0))                         # period too long: 419 s/150
        files.append(_write_trial(tmp_path, dm, rows)[0])
    sifting.long_period = 2.0
    cl = sifting.read_candidates(files, prelim_reject=True, track=True)
    assert len(cl.badlists["threshold"]) == 5 and len(cl.badlists["longperiod"]) == 5
    n0 = len(cl)
    cl = sifting.remove_duplicate_candidates(cl, verbosity=0)
    assert len(cl.duplicates) == n0 - len(cl)
    best = [c for c in cl if abs(c.r - P) < 1.0][0]
    assert best.DM == 5.0 and sorted(h[0] for h in best.hits) == [2.5, 5.0, 7.5]
    cl = sifting.remove_DM_problems(cl, 2, ["%.2f" % d for d in dms], 2.0, verbosity=0)
    kept_r = sorted(round(c.r) for c in cl)
    assert round(P) in kept_r and round(2 * P) in kept_r
    assert 60000 not in kept_r and 33333 not in kept_r and 45001 not in kept_r
    notes = " | ".join(c.note for c in cl.badlists["dmproblem"])
    assert "only 1 DM hits" in notes and "low DM cutoff" in notes and "gaps" in notes
    cl = sifting.remove_harmonics(cl, verbosity=0)
    assert [round(c.r) for c in cl] == [round(P)]
    assert "harmonic (2nd)" in cl.badlists["harmonic"][0].note
    out = str(tmp_path / "sifted.txt")
    cl.to_file(out)
    text = open(out).read()
    assert text.startswith("#") and "DM=  5.00" in text and text.count("\n  DM=") == 3
    cl.print_cand_summary(str(tmp_path / "summary.txt"))
    assert "Candlist contains 1 'good' candidates" in open(str(tmp_path / "summary.txt")).read()
    sifting.long_period = 15.0


def test_rogue_harmonic_and_power_cutoff(tmp_path):
    import presto_amd.api as api
    name, c, setup = _write_trial(tmp_path, 3.0, [(9000.5, 0.0, 8, 9.0, 6.0), (7000.5, 0.0, 8, 9.0, 3.0)])
    cl = sifting.candlist_from_candfile(name, trackbad=True)
    cl[0].harm_pows[6] = 40.0       # one high-numbered harmonic dominates
    cl.reject_harmpowcutoff()
    assert len(cl) == 1 and "All harmonics have power < 8" in cl.badlists["harmpowcutoff"][0].note
    cl.reject_rogueharmpow()
    assert len(cl) == 0 and "High-numbered harmonic (6)" in cl.badlists["rogueharmpow"][0].note


def _frozen():
    import importlib.util
    import json
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_sifting_frozen", os.path.join(here, "make_sifting_frozen.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with open(os.path.join(here, "sifting_frozen.json")) as f:
        return mod, json.load(f), here


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15])
def test_cross_dm_steps_reproduce_the_frozen_outputs(seed):
    """Duplicates, harmonics, DM problems and the whole sequence on fixed synthetic lists (clusters, chains that
    re-anchor on their strongest member, ties, integer and ratio harmonics, low-DM peaks, DM gaps): kept
    candidates in order, their hits, duplicates and rejects in order, every note -- as frozen from this module
    before its insides were rewritten (its own outputs: they pin nothing against the reference)."""
    import json
    mod, frozen, _ = _frozen()
    got = mod.run_case(seed)
    for step, desc in got.items():
        if str(seed) in frozen["cases"]:
            assert json.loads(json.dumps(desc)) == frozen["cases"][str(seed)][step], step
        assert mod.digest(desc) == frozen["case_digests"][str(seed)][step], step


def test_accel_file_parser_reproduces_the_frozen_outputs(tmp_path):
    import json
    mod, frozen, here = _frozen()
    import os
    small = mod.parsed_file(os.path.join(here, "post_small_DM7.50_ACCEL_30"))
    assert json.loads(json.dumps(small)) == frozen["post_small_parsed"]
    big = mod.parsed_file(mod.big_accel_file(str(tmp_path / "big_DM3.00_ACCEL_50")))     # candidate numbers past 9999
    assert len(big) == frozen["big_file"]["count"] and [b[0] for b in big[9997:10002]] == [9998, 9999, 10000, 10001, 10002]
    assert json.loads(json.dumps(big[9997:10002])) == frozen["big_file"]["around_9999"]
    assert mod.digest(big) == frozen["big_file"]["digest"]
