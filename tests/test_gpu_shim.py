"""The reference-signature shim (presto_amd/shim/accel_shim.c) driven through the
reference's own per-block call sequence must produce exactly what the whole-plane
entry points produce: same candidates, same work-file lines."""
import os
import subprocess

import numpy as np
import pytest

from presto_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shim_main_loop_matches_batched_search(tmp_path):
    import presto_amd as pa
    exe = os.path.join(ROOT, "presto_amd", "shim", "mainloop_test")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe), "-s"])
    N, dt, zmax, numharm = 1 << 17, 1e-3, 20, 8
    fft = synth.make_fft(N, dt, [dict(f=37.3, z=7.0, amp=0.35, duty=0.05)], seed=5)
    root = str(tmp_path / "shim")
    synth.write_fft(root, fft, N, dt)
    out = subprocess.check_output([exe, root + ".fft", str(N), repr(dt), str(zmax), str(numharm),
                                   root + ".txtcand"], timeout=170).decode()
    rows = [l.split()[1:] for l in out.splitlines() if l.startswith("CAND")]
    shim = np.array([[float(x) for x in r] for r in rows])

    g = pa.AccelSearchGPU(0)
    setup = g.configure(fft.size, N, dt, zmax, numharm, fhi=1e9)
    g.load_fft(fft)
    hits, cands, sig, added = g.search()
    g.close()
    assert len(cands) == len(shim) > 0
    assert np.array_equal(shim[:, 0].astype(np.float32), cands["power"])
    np.testing.assert_allclose(shim[:, 1], cands["sigma"], rtol=1e-6)
    assert np.array_equal(shim[:, 2].astype(int), cands["numharm"])
    assert np.array_equal(shim[:, 3], cands["r"]) and np.array_equal(shim[:, 4], cands["z"])
    # one work-file line per accepted insertion, in the same order, same format
    lines = open(root + ".txtcand").read().splitlines()
    acc = np.nonzero(added)[0]
    assert len(lines) == len(acc)
    T = N * dt
    for line, i in zip(lines, acc):
        nh = 1 << int(hits["stage"][i])
        rr = hits["rind"][i] * 0.5 / nh
        zz = (setup.zlo + hits["zind"][i] * 2.0) / nh
        want = "%-7.2f  %-7.4f  %-2d  %-14.4f  %-14.9f  %-10.4f" % (hits["power"][i], sig[i], nh, rr, rr / T, zz)
        assert line == want


def test_command_line_front_end(tmp_path):
    """accelsearch_hip on .fft/.inf files: same work-file lines as the library path."""
    import presto_amd as pa
    exe = os.path.join(ROOT, "presto_amd", "host", "accelsearch_hip")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe), "-s"])
    N, dt, zmax, numharm = 1 << 18, 5e-4, 51, 4          # odd zmax: searched as 52, files named 51 (src/accel_utils.c:1492-1494 before :1526)
    fft = synth.make_fft(N, dt, [dict(f=77.7, z=-31.0, amp=0.3, duty=0.05)], seed=8)
    root = str(tmp_path / "trial_DM12.30")
    synth.write_fft(root, fft, N, dt)
    out = subprocess.check_output([exe, "-zmax", str(zmax), "-numharm", str(numharm), "-flo", "2.0",
                                   "-fhi", "900", "-inmem", root + ".fft"], timeout=170).decode()
    assert "FFT length = 8192" in out and "Searched the following approx numbers" in out
    lines = open(root + "_ACCEL_51.txtcand").read().splitlines()
    g = pa.AccelSearchGPU(0)
    setup = g.configure(fft.size, N, dt, zmax, numharm, flo=2.0, fhi=900.0)
    g.load_fft(fft)
    hits, cands, sig, added = g.search()
    g.close()
    assert setup.zmax == 52 and len(lines) == int(added.sum()) > 0
    T = N * dt
    i = np.nonzero(added)[0][0]
    nh = 1 << int(hits["stage"][i])
    rr = hits["rind"][i] * 0.5 / nh
    zz = (setup.zlo + hits["zind"][i] * 2.0) / nh
    assert lines[0] == "%-7.2f  %-7.4f  %-2d  %-14.4f  %-14.9f  %-10.4f" % (hits["power"][i], sig[i], nh, rr, rr / T, zz)
    best = cands[np.argmax(cands["sigma"])]
    first = [l for l in out.splitlines() if l.startswith("sigma:")][0]
    assert "r: %-14.4f" % best["r"] in first
    # bad input is refused with the reference's message
    bad = subprocess.run([exe, root + ".txt"], capture_output=True, text=True)
    assert bad.returncode == 1 and "must be an '.fft' or '.[s]dat' file" in bad.stdout


def test_shim_main_loop_standard_mode(tmp_path):
    """The reference's standard-mode call sequence through the shim (subharm_ffdot_plane
    for every harmonic + add_ffdotpows) gives the candidates of accelgpu_run_standard."""
    import presto_amd as pa
    exe = os.path.join(ROOT, "presto_amd", "shim", "mainloop_test")
    N, dt, zmax, numharm = 1 << 17, 1e-3, 20, 8
    fft = synth.make_fft(N, dt, [dict(f=37.3, z=7.0, amp=0.35, duty=0.05)], seed=5)
    root = str(tmp_path / "shimstd")
    synth.write_fft(root, fft, N, dt)
    out = subprocess.check_output([exe, root + ".fft", str(N), repr(dt), str(zmax), str(numharm),
                                   root + ".txtcand", "0"], timeout=170).decode()
    assert "Harmonic  3/8" in out
    rows = [l.split()[1:] for l in out.splitlines() if l.startswith("CAND")]
    shim = np.array([[float(x) for x in r] for r in rows])
    g = pa.AccelSearchGPU(0)
    g.configure(fft.size, N, dt, zmax, numharm, fhi=1e9, inmem=False)
    g.load_fft(fft)
    hits, cands, sig, added = g.search_standard()
    g.close()
    assert len(cands) == len(shim) > 0
    assert np.array_equal(shim[:, 0].astype(np.float32), cands["power"])
    assert np.array_equal(shim[:, 2].astype(int), cands["numharm"])
    assert np.array_equal(shim[:, 3], cands["r"]) and np.array_equal(shim[:, 4], cands["z"])


def test_command_line_standard_mode(tmp_path):
    import presto_amd as pa
    exe = os.path.join(ROOT, "presto_amd", "host", "accelsearch_hip")
    N, dt, zmax, numharm = 1 << 18, 5e-4, 50, 8
    fft = synth.make_fft(N, dt, [dict(f=77.7, z=-31.0, amp=0.3, duty=0.05)], seed=8)
    root = str(tmp_path / "std_DM3.00")
    synth.write_fft(root, fft, N, dt)
    out = subprocess.check_output([exe, "-zmax", str(zmax), "-numharm", str(numharm), "-standard",
                                   root + ".fft"], timeout=170).decode()
    assert "using standard accelsearch" in out
    lines = open(root + "_ACCEL_50.txtcand").read().splitlines()
    g = pa.AccelSearchGPU(0)
    g.configure(fft.size, N, dt, zmax, numharm, inmem=False)
    g.load_fft(fft)
    hits, cands, sig, added = g.search_standard()
    g.close()
    assert len(lines) == int(added.sum()) > 0


def test_command_line_dat_input(tmp_path):
    """accelsearch_hip on a .dat time series: realfft + deredden on the GPU, same
    candidates as the library path; no work file for .dat input."""
    import presto_amd as pa
    exe = os.path.join(ROOT, "presto_amd", "host", "accelsearch_hip")
    N, dt, zmax, numharm = 1 << 19, 5e-4, 20, 4
    x = synth.pulsar_series(N, dt, [dict(f=41.7, z=3.0, amp=0.4, duty=0.05)], seed=11).astype(np.float32)
    root = str(tmp_path / "series_DM7.00")
    x.tofile(root + ".dat")
    open(root + ".inf", "w").write(synth.inf_text(os.path.basename(root), N, dt))
    out = subprocess.check_output([exe, "-zmax", str(zmax), "-numharm", str(numharm), root + ".dat"],
                                  timeout=170).decode()
    assert "Removing red-noise...done." in out
    assert not os.path.exists(root + "_ACCEL_20.txtcand")
    g = pa.AccelSearchGPU(0)
    g.configure(N // 2, N, dt, zmax, numharm)
    g.load_dat(x, deredden=True)
    hits, cands, sig, added = g.search()
    g.close()
    best = cands[np.argmax(cands["sigma"])]
    first = [l for l in out.splitlines() if l.startswith("sigma:")][0]
    assert "r: %-14.4f" % best["r"] in first and "harm: %-2d" % best["numharm"] in first
    assert abs(best["r"] / (N * dt) - 41.7) < 0.01 * 4


def test_command_line_zaplist_on_dat_input(tmp_path):
    """-zaplist/-baryv zap the spectrum made from a time series (src/accelsearch.c:82-105):
    the zapped tone disappears from the ACCEL file, the pulsar stays, and the candidates
    equal the library path with the same steps."""
    import presto_amd as pa
    import presto_amd.api as api
    from presto_amd import sifting
    exe = os.path.join(ROOT, "presto_amd", "host", "accelsearch_hip")
    N, dt, zmax, numharm = 1 << 19, 5e-4, 20, 4
    T = N * dt
    x = synth.pulsar_series(N, dt, [dict(f=41.7, z=3.0, amp=0.4, duty=0.05),
                                    dict(f=60.0, z=0.0, amp=0.8, duty=0.5)], seed=12).astype(np.float32)
    root = str(tmp_path / "zap_DM3.00")
    x.tofile(root + ".dat")
    open(root + ".inf", "w").write(synth.inf_text(os.path.basename(root), N, dt))
    zap = tmp_path / "mains.zaplist"
    zap.write_text("# freq width\n60.0 0.5\n120.0 0.5\n180.0 0.5\nB900.0 5000.0\n")   # the last one runs past Nyquist
    args = [exe, "-zmax", str(zmax), "-numharm", str(numharm), "-sigma", "4"]
    subprocess.check_output(args + [root + ".dat"], timeout=170)
    plain = sifting.candlist_from_candfile(root + "_ACCEL_20")
    assert any(abs(c.f - 60.0) < 0.05 for c in plain)
    out = subprocess.check_output(args + ["-zaplist", str(zap), "-baryv", "1e-4", root + ".dat"], timeout=170).decode()
    assert "Zapping them using a barycentric velocity of 0.0001c." in out
    zapped = sifting.candlist_from_candfile(root + "_ACCEL_20")
    assert not any(abs(c.f - 60.0 * k) < 0.2 for c in zapped for k in (1, 2, 3))
    assert any(abs(c.f - 41.7) < 0.02 for c in zapped)
    # library path, same steps
    g = pa.AccelSearchGPU(0)
    g.configure(N // 2, N, dt, zmax, numharm, sigma=4.0)
    g.load_dat(x, deredden=True)
    lo, hi = api.read_birds(str(zap), T, 1e-4)
    keep = lo < N // 2
    g.zapbirds(lo[keep], np.minimum(hi[keep], N // 2 - 1))
    hits, cands, sig, added = g.search()
    kept = api.sort_and_eliminate(cands)
    g.close()
    assert len(kept) == len(zapped)
    assert [c.numharm for c in zapped] == list(kept["numharm"])
    assert np.allclose([c.r for c in zapped], kept["r"], atol=0.26)


@pytest.mark.parametrize("inmem", [1, 0])
def test_shim_main_loop_otheropt_branches(tmp_path, inmem):
    """With -otheropt the reference's loop calls fund_to_ffdotplane_trans / inmem_add_ffdotpows_trans /
    add_ffdotpows_ptrs (src/accelsearch.c:154-155, 184-185, 193-195, 204-206): the same numbers from another
    layout.  Through the shim they must give exactly the candidates and the work file of the plain calls."""
    exe = os.path.join(ROOT, "presto_amd", "shim", "mainloop_test")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe), "-s"])
    N, dt, zmax, numharm = 1 << 17, 1e-3, 20, 8
    fft = synth.make_fft(N, dt, [dict(f=37.3, z=7.0, amp=0.35, duty=0.05)], seed=5)
    root = str(tmp_path / "shimopt")
    synth.write_fft(root, fft, N, dt)
    outs, works = [], []
    for otheropt in (0, 1):
        out = subprocess.check_output([exe, root + ".fft", str(N), repr(dt), str(zmax), str(numharm),
                                       root + ".txtcand", str(inmem), str(otheropt)], timeout=170).decode()
        outs.append([l for l in out.splitlines() if l.startswith("CAND")])
        works.append(open(root + ".txtcand").read())
    assert len(outs[0]) > 0 and outs[0] == outs[1] and works[0] == works[1]
