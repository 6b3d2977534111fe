"""Pins for the CPU oracle (no GPU).

Pinned against the reference's own code (the five self-contained files built
unmodified into oracle/_ref/libprestoref.so): Fresnel integrals, the quickselect
median, the chi^2 / normal routines behind power_for_sigma and candidate_sigma,
the downhill simplex of the candidate optimisation, and (two more unmodified files, the
generated parsers accelsearch_cmd.c and rednoise_cmd.c) the command line.  Pinned against mathematics: the z-response versus quadrature
of its defining integral, the FFT versus a double-precision DFT.  Checked against an
independent evaluation of the reference's set-up formulas (tests/golden/setup_tables.json,
computed with SciPy, not output of the reference program): half-widths, FFT lengths,
block counts, numindep and powcut tables.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def ref(oracle):
    r = oracle.ref_lib()
    if r is None:
        pytest.skip("oracle/_ref/libprestoref.so not built (reference tree absent)")
    return r


def test_fresnel_matches_reference(oracle, ref):
    L = oracle.lib()
    xs = np.concatenate([np.linspace(-60, 60, 4001), [0.0, 1.6, 1.6007, 36974.0, 36975.0, 1e-9],
                         np.random.default_rng(1).uniform(-400, 400, 2000)])
    for x in xs:
        s1, c1, s2, c2 = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        L.orc_fresnel(x, C.byref(s1), C.byref(c1))
        ref.fresnl(x, C.byref(s2), C.byref(c2))
        assert s1.value == s2.value and c1.value == c2.value, x


def test_median_matches_reference(oracle, ref):
    rng = np.random.default_rng(2)
    for n in (1, 2, 3, 4, 5, 10, 101, 3768, 3964, 4096, 5176):
        for rep in range(3):
            a = rng.exponential(1.0, n).astype(np.float32)
            if rep == 2:
                a[: n // 2] = a[0]            # heavy ties
            b = a.copy()
            got = oracle.median(a)
            want = float(ref.median(b.ctypes.data_as(C.c_void_p), n))
            assert got == want
            assert got == float(np.sort(a)[(n - 1) // 2])


def test_amoeba_matches_reference(oracle, ref):
    """The oracle's simplex beside the reference's amoeba() (src/amoeba.c:18-75) on the
    same functions, start simplexes and tolerances: the same vertices, values and
    evaluation counts, bit for bit, including a run that stops at the evaluation cap."""
    L = oracle.lib()
    FN = C.CFUNCTYPE(C.c_double, C.POINTER(C.c_double))
    P = (C.c_double * 2) * 3
    Y = C.c_double * 3
    for lib in (L.orc_amoeba, ref.amoeba):
        lib.argtypes = [C.POINTER(C.c_double * 2), C.POINTER(C.c_double), C.c_double, FN, C.POINTER(C.c_int)]
        lib.restype = None
    rng = np.random.default_rng(11)
    noise = rng.normal(0.0, 1e-9, 4096)

    def bowl(x):
        return (x[0] - 1.25) ** 2 + 3.0 * (x[1] + 0.5) ** 2 + 0.3 * (x[0] - 1.25) * (x[1] + 0.5)

    def rosen(x):
        return (1.0 - x[0]) ** 2 + 100.0 * (x[1] - x[0] * x[0]) ** 2

    def peak(x):      # a negative power-like peak, as the optimisation minimises
        return -50.0 * np.sinc(x[0] - 0.3) ** 2 * np.sinc(0.25 * (x[1] - 1.7)) ** 2

    def rough(x):     # round-off-like noise: never reaches the tolerance, stops at the cap
        k = int(abs(x[0] * 7919.0 + x[1] * 104729.0) * 1e6) % 4096
        return bowl(x) + noise[k]

    ncap = 0
    for f in (bowl, rosen, peak, rough):
        for ftol in (1e-7, 1e-10, 1e-15):
            for rep in range(4):
                x0, z0 = rng.uniform(-2, 2, 2)
                step = (0.4, 0.01, 1.0, 0.4)[rep]
                start = [[x0 - step, z0 - step], [x0 - step, z0 + step], [x0 + step, z0]]
                outs = []
                for lib in (L.orc_amoeba, ref.amoeba):
                    calls = []

                    def cb(px, f=f, calls=calls):
                        v = float(f((px[0], px[1])))
                        calls.append((px[0], px[1]))
                        return v
                    p = P(*[(C.c_double * 2)(*v) for v in start])
                    y = Y(*[float(f(v)) for v in start])
                    n = C.c_int(-1)
                    lib(p, y, ftol, FN(cb), C.byref(n))
                    outs.append(([list(r) for r in p], list(y), n.value, calls))
                assert outs[0] == outs[1], (f.__name__, ftol, rep)
                ncap += outs[0][2] >= 5000
    assert ncap > 0          # the evaluation cap was exercised


def _ref_power_for_sigma(ref, sigma, numsum, numtrials):
    d, i = C.c_double, C.c_int
    which, status = i(1), i(0)
    p, q, x, bound, mean, sd = d(), d(), d(sigma), d(), d(0.0), d(1.0)
    ref.cdfnor(C.byref(which), C.byref(p), C.byref(q), C.byref(x), C.byref(mean), C.byref(sd),
               C.byref(status), C.byref(bound))
    assert status.value == 0
    q = d(q.value / numtrials)
    p = d(1.0 - q.value)
    which, status, df = i(2), i(0), d(2.0 * numsum)
    ref.cdfchi(C.byref(which), C.byref(p), C.byref(q), C.byref(x), C.byref(df), C.byref(status),
               C.byref(bound))
    assert status.value == 0
    return 0.5 * x.value


def _ref_candidate_sigma(ref, power, numsum, numtrials):
    """candidate_sigma() of the reference for the non-asymptotic branch."""
    d, i = C.c_double, C.c_int
    which, status = i(1), i(0)
    p, q, x, df, bound = d(), d(), d(2.0 * power), d(2.0 * numsum), d()
    ref.cdfchi(C.byref(which), C.byref(p), C.byref(q), C.byref(x), C.byref(df), C.byref(status),
               C.byref(bound))
    assert status.value == 0
    logp = np.log(q.value) + np.log(numtrials)
    qq = np.exp(logp)
    pp = 1.0 - qq
    if pp <= 0:
        return 0.0
    which, status = i(2), i(0)
    p, q, x, mean, sd = d(pp), d(qq), d(), d(0.0), d(1.0)
    ref.cdfnor(C.byref(which), C.byref(p), C.byref(q), C.byref(x), C.byref(mean), C.byref(sd),
               C.byref(status), C.byref(bound))
    assert status.value == 0
    return max(x.value, 0.0)


def test_power_for_sigma_matches_reference(oracle, ref):
    L = oracle.lib()
    nbad_float = 0
    cases = [(s, nh, nt) for s in (1.0, 2.0, 3.5, 6.0) for nh in (1, 2, 4, 8, 16, 32)
             for nt in (1e3, 524183.0, 15236347.0, 243781563.0, 5831448272.0)]
    for s, nh, nt in cases:
        got = L.orc_power_for_sigma(s, nh, nt)
        want = _ref_power_for_sigma(ref, s, nh, nt)
        assert abs(got - want) <= 2e-8 * want, (s, nh, nt, got, want)   # DCDFLIB stops at 1e-8
        nbad_float += np.float32(got) != np.float32(want)
    # the float powcut the search uses is the same value in (nearly) every case
    assert nbad_float <= len(cases) // 10


def test_candidate_sigma_matches_reference(oracle, ref):
    L = oracle.lib()
    rng = np.random.default_rng(3)
    for nh in (1, 2, 4, 8, 16, 32):
        for nt in (65522.0, 15236347.0, 2901241976.0):
            for power in np.concatenate([rng.uniform(nh * 0.2, nh * 14.5, 40), [nh * 1.0]]):
                got = L.orc_candidate_sigma(float(power), nh, nt)
                want = _ref_candidate_sigma(ref, float(power), nh, nt)
                assert abs(got - want) <= 1e-9 * max(1.0, want), (power, nh, nt, got, want)


def test_candidate_sigma_asymptotic_branch_is_continuous(oracle):
    # chi2/dof > 15 switches to the A&S expansions (src/characteristics.c:504-509):
    # the two branches must agree where they meet, and sigma must keep rising.
    L = oracle.lib()
    for nh in (1, 4, 16):
        lo = L.orc_candidate_sigma(nh * 14.999, nh, 1e6)
        hi = L.orc_candidate_sigma(nh * 15.001, nh, 1e6)
        assert 0 <= hi - lo < 2e-2
        big = [L.orc_candidate_sigma(nh * f, nh, 1e6) for f in (20, 50, 200, 1000)]
        assert all(b > a for a, b in zip(big, big[1:]))


def test_z_response_is_the_chirp_integral(oracle):
    L = oracle.lib()
    u = (np.arange(40000) + 0.5) / 40000
    for z in (2.0, -2.0, 37.0, -50.0, 200.0, -200.0, 300.0, 1200.0, -1200.0):
        m = L.orc_z_resp_halfwidth(z)
        nk = 4 * m
        k = oracle.gen_z_response(z, nk)
        r = (-0.5 * z + nk / 4.0) - np.arange(nk) * 0.5
        idx = np.unique(np.concatenate([np.arange(0, nk, max(nk // 60, 1)), [nk // 2, nk - 1]]))
        ph = np.outer(r[idx], u) + 0.5 * z * u * u
        want = np.exp(2j * np.pi * ph).mean(axis=1)
        assert np.abs(want - k[idx]).max() < 2e-6, z
    # z = 0 falls through to the sinc kernel
    k0 = oracle.gen_z_response(0.0, 64)
    r = (64 / 4.0) - np.arange(64) * 0.5
    want = np.where(r == 0, 1.0, np.sin(np.pi * r) / np.where(r == 0, 1, np.pi * r)) * np.exp(1j * np.pi * r)
    assert np.abs(k0 - want).max() < 1e-6


def test_fft_against_double_dft(oracle):
    rng = np.random.default_rng(4)
    for n in (2, 4, 8, 64, 128, 4096, 8192, 16384):
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        f = np.fft.fft(x.astype(np.complex128))
        b = np.fft.ifft(x.astype(np.complex128)) * n
        assert np.abs(oracle.fft(x, -1) - f).max() <= 4e-7 * np.abs(f).max() * np.log2(n)
        assert np.abs(oracle.fft(x, +1) - b).max() <= 4e-7 * np.abs(b).max() * np.log2(n)


def test_setup_tables(oracle):
    """Shapes and thresholds recorded from the reference program (golden file)."""
    L = oracle.lib()
    g = json.load(open(os.path.join(HERE, "golden", "setup_tables.json")))
    for z, m in g["halfwidth"].items():
        assert L.orc_z_resp_halfwidth(float(z)) == m
    for z, n in g["fftlen"].items():
        assert L.orc_calc_fftlen(1, 1, int(z)) == n
    for c in g["configs"]:
        fft = np.zeros(16, dtype=np.complex64)          # never dereferenced here
        s = oracle.Search.__new__(oracle.Search)
        s.fft = fft
        s.obs = oracle.Obs()
        rc = L.orc_obs_init(C.byref(s.obs), fft.ctypes.data_as(C.c_void_p), c["numbins"],
                            2 * c["numbins"], c["dt"], c["zmax"], c["numharm"], 2.0, 1.0, 1e9,
                            0, 0, 0, 0)
        assert rc == 0
        assert s.obs.rlo == c["rlo"] and s.obs.numz == c["numz"] and s.obs.fftlen == c["fftlen"]
        assert s.num_blocks == c["blocks"] and s.num_prepop == c["prepop"]
        assert abs(s.obs.rlen * s.obs.numz * 4 / 1e9 - c["plane_gb"]) <= 1.2e-2 * c["plane_gb"]
        assert list(s.numindep) == c["numindep"]
        np.testing.assert_allclose(s.powcut, c["powcut"], rtol=2e-5)   # table has 5 digits


def test_index_helpers(oracle):
    L = oracle.lib()
    # round-half-away-from-zero on the z grid, truncation on the r grid
    assert L.orc_calc_required_z(0.5, 2.0) == 2 and L.orc_calc_required_z(0.5, -2.0) == -2
    assert L.orc_calc_required_z(0.25, 2.0) == 0 and L.orc_calc_required_z(0.75, -6.0) == -4
    assert L.orc_calc_required_r(0.5, 1001.5) == 501.0
    assert L.orc_calc_required_r(1.0, 536.0) == 536.0
    assert L.orc_index_from_z(-4.0, -200.0) == 98


def test_realfft_against_double_precision_transform(oracle):
    """orc_realfft_forward (reference src/fastffts.c:198-270): packed layout, sign and scale."""
    rng = np.random.default_rng(4)
    for n in (1 << 10, 1 << 15):
        x = rng.standard_normal(n).astype(np.float32)
        got = oracle.realfft(x)
        ref = np.fft.rfft(x.astype(np.float64))
        assert abs(got[0].real - ref[0].real) < 1e-3 and abs(got[0].imag - ref[-1].real) < 1e-3
        assert np.abs(got[1:] - ref[1:-1]).max() < 2e-6 * np.abs(ref).max() * np.log2(n)


def test_deredden_whitens_and_is_deterministic(oracle):
    """orc_deredden (reference src/accel_utils.c:1301-1374): bin 0 becomes 1+0j, a red
    spectrum comes out with median power ln 2 everywhere, and the first three bins
    carry the plain scale of the first 6-bin block."""
    rng = np.random.default_rng(5)
    n = 1 << 16
    amp = 1.0 + 300.0 / (1.0 + np.arange(n) / 50.0)
    spec = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * amp).astype(np.complex64)
    out = oracle.deredden(spec)
    assert out[0] == 1.0
    pw = np.abs(out[1:]) ** 2
    for lo, hi in ((1, 2000), (2000, 20000), (20000, n - 1)):
        assert abs(np.median(pw[lo:hi]) / np.log(2.0) - 1.0) < 0.08
    p = (spec.real ** 2 + spec.imag ** 2).astype(np.float32)
    med = np.sort(p[1:7])[2]
    x = np.float32(np.float64(med) / np.log(2.0))
    i = np.array([x]).view(np.int32)[0]
    y = np.array([np.int32(0x5f3759df) - (i >> 1)], dtype=np.int32).view(np.float32)[0]
    nrm = np.float32(y * np.float32(np.float32(1.5) - np.float32(np.float32(np.float32(np.float32(0.5) * x) * y) * y)))
    assert out[1].real == np.float32(spec[1].real * nrm) and out[3].imag == np.float32(spec[3].imag * nrm)
    assert np.array_equal(out, oracle.deredden(spec))


# ---------------------------------------------------------------------------
# Command-line surface: the product's accelsearch_hip against the reference's own generated
# parser (src/accelsearch_cmd.c, compiled unmodified into oracle/_ref/libaccelsearch_cmd.so).
# The reference parser exit()s on errors, so it runs in a child process.
# ---------------------------------------------------------------------------
_REF_CLI = r'''
import ctypes as C, json, sys
class Cmdline(C.Structure):           # include/accelsearch_cmd.h:11-75
    _fields_ = [("ncpusP", C.c_char), ("ncpus", C.c_int), ("ncpusC", C.c_int),
                ("lobinP", C.c_char), ("lobin", C.c_int), ("lobinC", C.c_int),
                ("numharmP", C.c_char), ("numharm", C.c_int), ("numharmC", C.c_int),
                ("zmaxP", C.c_char), ("zmax", C.c_int), ("zmaxC", C.c_int),
                ("sigmaP", C.c_char), ("sigma", C.c_float), ("sigmaC", C.c_int),
                ("rloP", C.c_char), ("rlo", C.c_double), ("rloC", C.c_int),
                ("rhiP", C.c_char), ("rhi", C.c_double), ("rhiC", C.c_int),
                ("floP", C.c_char), ("flo", C.c_double), ("floC", C.c_int),
                ("fhiP", C.c_char), ("fhi", C.c_double), ("fhiC", C.c_int),
                ("inmemP", C.c_char), ("photonP", C.c_char), ("medianP", C.c_char), ("locpowP", C.c_char),
                ("zaplistP", C.c_char), ("zaplist", C.c_char_p), ("zaplistC", C.c_int),
                ("baryvP", C.c_char), ("baryv", C.c_double), ("baryvC", C.c_int),
                ("otheroptP", C.c_char), ("noharmpolishP", C.c_char), ("noharmremoveP", C.c_char),
                ("argc", C.c_int), ("argv", C.POINTER(C.c_char_p)), ("full_cmd_line", C.c_char_p)]
lib = C.CDLL(sys.argv[1])
lib.parseCmdline.restype = C.POINTER(Cmdline)
args = [b"accelsearch"] + [a.encode() for a in sys.argv[2:]]
argv = (C.c_char_p * (len(args) + 1))(*args, None)
c = lib.parseCmdline(len(args), argv).contents
b = lambda x: int(x != b"\x00")
print(json.dumps(dict(ncpus=c.ncpus, lobin=c.lobin, numharm=c.numharm, zmax=c.zmax, sigma=c.sigma, rlo=c.rlo, rhi=c.rhi,
                      flo=c.flo, fhi=c.fhi, floP=b(c.floP), fhiP=b(c.fhiP), rloP=b(c.rloP), rhiP=b(c.rhiP),
                      inmem=b(c.inmemP), photon=b(c.photonP), median=b(c.medianP), locpow=b(c.locpowP),
                      zaplist=c.zaplist.decode() if c.zaplist else "-", baryv=c.baryv, otheropt=b(c.otheroptP),
                      noharmpolish=b(c.noharmpolishP), noharmremove=b(c.noharmremoveP),
                      file=c.argv[0].decode() if c.argc == 1 else None)))
'''


def _ref_cli(args):
    import subprocess
    import sys
    so = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "libaccelsearch_cmd.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libaccelsearch_cmd.so not built (reference tree absent)")
    r = subprocess.run([sys.executable, "-c", _REF_CLI, so] + args, capture_output=True, timeout=60)
    return r.returncode, (json.loads(r.stdout.decode().strip().splitlines()[-1]) if r.returncode == 0 else None)


def _our_cli(args):
    import subprocess
    exe = os.path.join(os.path.dirname(HERE), "presto_amd", "host", "accelsearch_hip")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe), "-s"])
    r = subprocess.run([exe, "-showopts"] + args, capture_output=True, timeout=60)
    if r.returncode != 0:
        return r.returncode, None
    kv = dict(tok.split("=", 1) for tok in r.stdout.decode().split())
    return 0, kv


@pytest.mark.parametrize("args", [
    ["x.fft"],
    ["-zmax", "51", "-numharm", "16", "-sigma", "3.5", "-flo", "2", "-fhi", "900.5", "x.fft"],
    ["-rlo", "5", "-rhi", "70000", "-lobin", "1000", "-inmem", "-ncpus", "4", "y_DM12.30.fft"],
    ["-photon", "-noharmpolish", "-noharmremove", "-otheropt", "-zmax", "0", "-numharm", "1", "z.dat"],
    ["-locpow", "-median", "-zaplist", "birds.zaplist", "-baryv", "-0.0001", "-sigma", "30", "-zmax", "1200",
     "-numharm", "32", "t.sdat"],
])
def test_command_line_matches_the_reference_parser(args):
    rc_r, want = _ref_cli(args)
    rc_o, got = _our_cli(args)
    assert rc_r == 0 and rc_o == 0
    for k in ("ncpus", "lobin", "numharm", "zmax", "inmem", "photon", "median", "locpow", "otheropt", "noharmpolish",
              "noharmremove"):
        assert int(got[k]) == int(want[k]), k
    for k in ("sigma", "rlo", "rhi", "flo", "fhi", "baryv"):
        assert float(got[k]) == float(want[k]), k
    assert got["zaplist"] == want["zaplist"] and got["file"] == want["file"]
    # -flo and -fhi count as given whatever the command line says (defaults 1 Hz and 10 kHz):
    # the reason -rlo/-rhi never take effect (src/accel_utils.c:1581-1624)
    assert want["floP"] == 1 and want["fhiP"] == 1


@pytest.mark.parametrize("args", [
    ["-zmax", "1201", "x.fft"], ["-zmax", "-2", "x.fft"], ["-sigma", "0.5", "x.fft"], ["-sigma", "31", "x.fft"],
    ["-numharm", "64", "x.fft"], ["-numharm", "0", "x.fft"], ["-baryv", "0.2", "x.fft"], ["-lobin", "-1", "x.fft"],
    ["-flo", "-1", "x.fft"], ["-nosuchflag", "x.fft"], [], ["a.fft", "b.fft"], ["-zmax"],
])
def test_command_line_rejects_what_the_reference_parser_rejects(args):
    rc_r, _ = _ref_cli(args)
    rc_o, _ = _our_cli(args)
    assert rc_r != 0 and rc_o != 0


def test_rednoise_defaults_match_the_reference_parser():
    """startwidth 6, endwidth 100, endfreq 6.0: src/rednoise_cmd.c (clig/rednoise_cmd.cli:13-21),
    compiled unmodified, against the defaults of the host layer's rednoise()."""
    import inspect
    import subprocess
    import sys
    import presto_amd.api as api
    so = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "librednoise_cmd.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/librednoise_cmd.so not built (reference tree absent)")
    code = r'''
import ctypes as C, sys
class Cmdline(C.Structure):           # include/rednoise_cmd.h
    _fields_ = [("startwidthP", C.c_char), ("startwidth", C.c_int), ("startwidthC", C.c_int),
                ("endwidthP", C.c_char), ("endwidth", C.c_int), ("endwidthC", C.c_int),
                ("endfreqP", C.c_char), ("endfreq", C.c_double), ("endfreqC", C.c_int),
                ("argc", C.c_int), ("argv", C.POINTER(C.c_char_p)), ("full_cmd_line", C.c_char_p)]
lib = C.CDLL(sys.argv[1])
lib.parseCmdline.restype = C.POINTER(Cmdline)
args = [b"rednoise", b"x.fft"]
argv = (C.c_char_p * 3)(*args, None)
c = lib.parseCmdline(2, argv).contents
print(c.startwidth, c.endwidth, c.endfreq)
'''
    out = subprocess.check_output([sys.executable, "-c", code, so], timeout=60).decode().split()
    sig = inspect.signature(api.AccelSearchGPU.rednoise).parameters
    assert (int(out[0]), int(out[1]), float(out[2])) == (sig["startwidth"].default, sig["endwidth"].default,
                                                         sig["endfreq"].default) == (6, 100, 6.0)


@pytest.mark.parametrize("args", [
    [], ["-zmax", "300", "-numharm", "16", "-sigma", "4", "-flo", "0.5", "-fhi", "2000"],
    ["-locpow", "-noharmpolish", "-noharmremove", "-baryv", "0.00003", "-zaplist", "b.zaplist"],
])
def test_batch_driver_search_flags_match_the_reference_parser(args):
    """presto_amd.pipeline takes accelsearch's search flags under their own names, with the
    reference parser's defaults."""
    from presto_amd import pipeline
    rc, want = _ref_cli(args + ["x.fft"])
    assert rc == 0
    got = pipeline.parse_args(args + ["x.fft"])
    assert (got.zmax, got.numharm, got.sigma, got.flo, got.fhi, got.baryv) == \
        (want["zmax"], want["numharm"], want["sigma"], want["flo"], want["fhi"], want["baryv"])
    assert (int(got.locpow), int(got.noharmpolish), int(got.noharmremove)) == \
        (want["locpow"], want["noharmpolish"], want["noharmremove"])
    assert (got.zaplist or "-") == want["zaplist"]


@pytest.mark.parametrize("zmax,z,r", [(0, 0, 9000.0), (50, 30, 9000.5), (200, -200, 12000.0), (200, 126, 7777.5),
                                      (300, 300, 9000.0), (1200, -1100, 11000.5)])
def test_matched_template_recovers_the_tone_power(oracle, zmax, z, r):
    """Known answer from the reference's own demonstration (python/ffdot_example.py:8-26:
    "Fraction of recovered signal power" of a noiseless drifting tone against N^2/4): the F-Fdot
    cell at the tone's (r, z) holds the whole power, up to what the truncated templates
    (z_resp_halfwidth, LOWACC) leave out, and no other cell of the block holds more."""
    from presto_amd import synth
    N = 1 << 16
    fft = synth.chirp_fft(N, r, z)
    fft[0] = 1.0                                  # -photon with one "photon": powers stay raw
    s = oracle.Search(fft, N, 1e-3, zmax, 1, flo=0.0, fhi=1e9, photon=True, alloc_plane=False)
    startr = s.rlo + 3735.0 * int((r - s.rlo) // 3735.0)
    blk, rlo = s.fund_block(startr)
    zi, col = (z + zmax) // 2, int(round(2 * (r - rlo)))
    frac = blk[zi, col] / (N * N / 4.0)
    assert 0.97 <= frac <= 1.002, frac          # 0.979 at half-bin offsets, 0.995-1.0 on the grid
    assert blk.max() <= blk[zi, col] * (1 + 1e-6)
    s.close()


def test_search_constants_match_the_reference_header():
    """ACCEL_USELEN, ACCEL_NUMBETWEEN, ACCEL_DR, ACCEL_DZ, ACCEL_CLOSEST_R as include/accel.h:13-31
    defines them, read from the header itself when the reference tree is present."""
    import re
    hdr = "/root/reference/include/accel.h"
    if not os.path.exists(hdr):
        pytest.skip("reference tree absent")
    text = open(hdr).read()
    val = lambda name: float(re.search(r"^#define\s+%s\s+([0-9.]+)" % name, text, re.M).group(1))
    import presto_amd.api as api
    ours = open(os.path.join(os.path.dirname(HERE), "include", "accelgpu.h")).read()
    assert int(val("ACCEL_USELEN")) == api.USELEN == int(re.search(r"#define\s+ACCELGPU_USELEN\s+(\d+)", ours).group(1))
    orc = open(os.path.join(os.path.dirname(HERE), "oracle", "accel_oracle.h")).read()
    for ref_name, orc_name in (("ACCEL_USELEN", "ORC_USELEN"), ("ACCEL_NUMBETWEEN", "ORC_NUMBETWEEN"),
                               ("ACCEL_DR", "ORC_DR"), ("ACCEL_RDR", "ORC_RDR"), ("ACCEL_DZ", "ORC_DZ"),
                               ("ACCEL_RDZ", "ORC_RDZ"), ("ACCEL_CLOSEST_R", "ORC_CLOSEST_R")):
        assert val(ref_name) == float(re.search(r"#define\s+%s\s+([0-9.]+)" % orc_name, orc).group(1)), ref_name
    # the product's grid follows from them: columns are half bins (DR), rows step by DZ = 2
    s = api.plan(1 << 19, 1 << 20, 1e-4, 20, 2, fhi=1e9)
    assert s.numz == 2 * (20 // int(val("ACCEL_DZ"))) + 1 and s.rlen == (s.highestbin + api.USELEN) * int(val("ACCEL_RDR"))
