"""Host-side result files (no GPU): value(error) strings, derived properties and the
ACCEL / .cand writers of the product library against the oracle's restatement of
src/output.c and src/accel_utils.c:565-835, and against what the consumer parses
(lib/python/sifting.py:40-42, 1100-1198)."""
import os
import re

import numpy as np
import pytest

from presto_amd import synth


def test_library_exports_output_symbols():
    import presto_amd as pa
    L = pa.load_library()
    for name in ("accelgpu_nice_output", "accelgpu_calc_props", "accelgpu_write_accel_files",
                 "accelgpu_sort_and_eliminate", "accelgpu_optimize_cands"):
        assert hasattr(L, name)


def test_nice_output_known_strings():
    """Hand-checked cases of the journal notation."""
    import presto_amd.api as api
    assert api.nice_output(1, 12945.375, 0.03) == " 12945.38(3)"
    assert api.nice_output(2, 12945.375, 0.03125) == " 12945.375(31)"
    assert api.nice_output(1, 1.2345678, 0.0002) == " 1.2346(2)"
    assert api.nice_output(2, -31.25, 0.5) == "-31.25(50)"
    assert api.nice_output(2, 0.0, 2.0) == " 0.0(2.0)"
    assert api.nice_output(1, 123456.0, 200.0) == " 1.235(2)x10^5"


def test_nice_output_matches_oracle_over_a_sweep(oracle):
    import presto_amd as pa
    import presto_amd.api as api
    rng = np.random.default_rng(0)
    vals = np.concatenate([[0.0, 1.0, -1.0, 10.0, 0.1, 99.95, 1e-7, -3e-9, 1e9, 12345.678],
                           rng.standard_normal(300) * 10.0 ** rng.integers(-9, 9, 300)])
    errs = np.concatenate([[0.0, 1.0, 0.095, 0.95, 9.5, 1e-5, 2e-3, 50.0, 1e3, 0.5],
                           np.abs(rng.standard_normal(300)) * 10.0 ** rng.integers(-9, 6, 300)])
    n1 = n2 = 0
    for v in vals[:60]:
        for e in errs[:60]:
            for digits in (1, 2):
                want = oracle.nice_output(digits, v, e)
                if want is None:              # the reference's "undefined condition": an error here
                    with pytest.raises(pa.AccelGPUError):
                        api.nice_output(digits, v, e)
                    continue
                assert api.nice_output(digits, v, e) == want, (digits, v, e)
                n1 += digits == 1
                n2 += digits == 2
    assert n1 > 3000 and n2 > 2500


def _fake_optcands(oracle, n, seed, T):
    rng = np.random.default_rng(seed)
    c = np.zeros(n, dtype=oracle.OPTCAND_DTYPE)
    c["numharm"] = 2 ** rng.integers(0, 5, n)
    c["r"] = np.sort(rng.uniform(200, 2e6, n))
    c["r"][1] = c["r"][0] * 2 + 0.01            # a harmonic pair and a side-lobe pair for the notes
    c["r"][3] = c["r"][2] + 4.0
    c["z"] = rng.integers(-100, 100, n) * 2.0 / c["numharm"]
    c["z"][1] = c["z"][0] * 2
    c["z"][3] = c["z"][2] + 6.0
    c["power"] = rng.uniform(20, 400, n)
    c["sigma"] = rng.uniform(2, 40, n)
    for i in range(n):
        nh = int(c["numharm"][i])
        c["hirs"][i, :nh] = (c["r"][i] + rng.uniform(-0.2, 0.2)) * np.arange(1, nh + 1)
        c["hizs"][i, :nh] = (c["z"][i] + rng.uniform(-0.5, 0.5)) * np.arange(1, nh + 1)
        d = c["derivs"][i]
        d["locpow"][:nh] = rng.uniform(0.5, 2.0, nh)
        d["pow"][:nh] = d["locpow"][:nh] * rng.uniform(0.5, 80, nh)
        d["phs"][:nh] = rng.uniform(0, 2 * np.pi, nh)
        d["dphs"][:nh] = rng.uniform(-3, 3, nh)
        d["dpow"][:nh] = rng.standard_normal(nh)
        d["d2pow"][:nh] = -d["pow"][:nh] * rng.uniform(0.5, 8, nh)
        c["pows"][i, :nh] = d["pow"][:nh]
    c["derivs"]["pow"][3, 0] = c["derivs"]["pow"][2, 0] * 0.1 * c["derivs"]["locpow"][3, 0] / c["derivs"]["locpow"][2, 0]
    return c


@pytest.mark.parametrize("photon", [False, True])
def test_accel_files_match_oracle_byte_for_byte(oracle, tmp_path, photon):
    import presto_amd as pa
    import presto_amd.api as api
    N, dt = 1 << 23, 64e-6
    fft = np.zeros(N // 2, dtype=np.complex64)
    fft[0] = 1234.0
    s = oracle.Search(fft, N, dt, 200, 16, fhi=1e9, alloc_plane=False, photon=photon)
    setup = pa.plan(N // 2, N, dt, 200, 16, fhi=1e9)
    c = _fake_optcands(oracle, 40, 7, N * dt)
    inf = synth.inf_text("fake_DM12.50", N, dt)
    a1, c1 = str(tmp_path / "o_ACCEL_200"), str(tmp_path / "o_ACCEL_200.cand")
    a2, c2 = str(tmp_path / "g_DM12.50_ACCEL_200"), str(tmp_path / "g_DM12.50_ACCEL_200.cand")
    s.write_accel_files(c, a1, c1, inf)
    g = np.zeros(len(c), dtype=api.OPTCAND_DTYPE)
    for k in c.dtype.names:
        g[k] = c[k]
    api.write_accel_files(setup, g, a2, c2, inf, nph=1234.0 if photon else 0.0)
    t1, t2 = open(a1).read(), open(a2).read()
    assert t1 == t2
    assert open(c1, "rb").read() == open(c2, "rb").read()
    assert os.path.getsize(c2) == 88 * len(c)
    assert ("NumPhot" in t2) == photon
    assert "H 2 of Cand 1" in t2 and "SL? of Cand 3" in t2
    props = np.fromfile(c2, dtype=api.FOURIERPROPS_DTYPE)
    assert np.array_equal(props["r"], c["r"]) and np.allclose(props["rerr"], 0.5 / c["numharm"])

    # what the consumer does with the file (lib/python/sifting.py:1100-1198)
    numsamp = tobs_dt = None
    for line in t2.splitlines():
        if line.startswith(" Number of bins in the time series"):
            numsamp = int(line.split()[-1])
        if line.startswith(" Width of each time series bin (sec)"):
            tobs_dt = float(line.split()[-1])
    assert numsamp == N and tobs_dt == dt
    assert re.search(r"DM(\d+\.\d{2})", a2).group(1) == "12.50"
    funds = [l.split() for l in t2.splitlines() if re.match(r"^\d", l)]
    harms = [l.split() for l in t2.splitlines() if re.match(r"^[ ]\d", l)]
    assert len(funds) == len(c) == len(harms)
    for i, (fl, hl) in enumerate(zip(funds, harms)):
        assert int(fl[0]) == i + 1 == int(hl[0])
        assert abs(float(fl[1]) - c["sigma"][i]) < 0.006 and int(fl[4]) == c["numharm"][i]
        assert abs(float(fl[7].split("(")[0]) - c["r"][i]) <= 0.5
        assert abs(float(fl[9].split("(")[0]) - c["z"][i]) <= 2.0
        float(hl[3].split("(")[0]); float(hl[9].split("(")[0])
    assert sum(1 for l in t2.splitlines() if re.match(r"^\s{5,}\d", l)) == int(c["numharm"].sum()) - len(c)
