"""The batch driver end to end on one GPU: DM-trial files (.fft and .dat) -> zap ->
search -> optimise -> ACCEL files -> sifting.  The injected pulsar survives with hits
at the neighbouring DMs; strong interference present only at DM 0 and a zapped tone
do not."""
import os

import numpy as np
import pytest

from presto_amd import pipeline, sifting, synth

pytestmark = pytest.mark.gpu


def test_pipeline_finds_the_pulsar_across_dm_trials(tmp_path):
    N, dt = 1 << 19, 2.5e-4
    T = N * dt
    files = []
    for k, dm in enumerate([0.0, 10.0, 20.0, 30.0]):
        puls = [dict(f=333.33, z=0.0, amp=0.6, duty=0.5)]                    # a "birdie" in every trial
        if dm in (10.0, 20.0, 30.0):
            puls.append(dict(f=77.7, z=-11.0, amp=0.30 - 0.004 * abs(dm - 20.0), duty=0.05))
        if dm == 0.0:
            puls.append(dict(f=50.0, z=0.0, amp=0.5, duty=0.3))              # terrestrial: DM 0 only
        root = str(tmp_path / ("beam_DM%.2f" % dm))
        if k == 3:
            x = synth.pulsar_series(N, dt, puls, seed=40 + k).astype(np.float32)
            x.tofile(root + ".dat")
            open(root + ".inf", "w").write(synth.inf_text(os.path.basename(root), N, dt))
            files.append(root + ".dat")
        else:
            synth.write_fft(root, synth.make_fft(N, dt, puls, seed=40 + k), N, dt)
            files.append(root + ".fft")
    zap = tmp_path / "birds.zaplist"
    zap.write_text("# freq width\n333.33 0.2\n666.66 0.3\n999.99 0.4\n1333.32 0.5\n")
    sifting.sigma_threshold, sifting.r_err = 6.0, 1.1
    opts = pipeline.parse_args(["-zmax", "20", "-numharm", "8", "-sigma", "3.0", "-fhi", "1900",
                                "-zaplist", str(zap), "-low_dm_cutoff", "2.0", "-verbosity", "0",
                                "-o", str(tmp_path / "beam")] + files)
    cl, secs = pipeline.run(sorted(files), opts)
    assert len(secs) == 4
    for dm in (0.0, 10.0, 20.0):
        assert os.path.exists(str(tmp_path / ("beam_DM%.2f_ACCEL_20" % dm)))
        assert os.path.exists(str(tmp_path / ("beam_DM%.2f_ACCEL_20.txtcand" % dm)))
    assert not os.path.exists(str(tmp_path / "beam_DM30.00_ACCEL_20.txtcand"))     # .dat input: no work file
    assert len(cl) >= 1
    best = cl[0]
    assert abs(best.f - 77.7) < 0.02 and abs(best.z + 11.0) < 1.5 and best.DM == 20.0
    assert sorted(h[0] for h in best.hits) == [10.0, 20.0, 30.0]
    for c in cl:
        assert abs(c.f - 50.0) > 0.05                       # DM-0 interference rejected
        for k in (1, 2, 3, 4):
            assert abs(c.f - 333.33 * k) > 0.3              # zapped tone and its harmonics gone
    text = open(str(tmp_path / "beam.accelcands")).read()
    assert "DM= 20.00" in text and text.count("\n  DM=") >= 3
    # the same files through the file-based route give the same sifted list
    accel = sorted(str(p) for p in tmp_path.glob("*_ACCEL_20"))
    cl2 = sifting.read_candidates(accel, prelim_reject=False, track=True)
    cl2 = pipeline.sift(cl2, ["0.00", "10.00", "20.00", "30.00"], 2, 2.0, verbosity=0)
    assert [(c.filename, c.candnum) for c in cl2] == [(c.filename, c.candnum) for c in cl]
    assert [c.sigma for c in cl2] == [c.sigma for c in cl]


def test_native_batch_program_equals_one_run_per_file(tmp_path):
    """accelbatch_hip (many files through one context, pipelined) writes byte for byte the
    ACCEL, .cand and .txtcand files that accelsearch_hip writes one file at a time,
    including a time-series input and a change of series length in mid-batch."""
    import shutil
    import subprocess
    root_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    one = os.path.join(root_dir, "presto_amd", "host", "accelsearch_hip")
    many = os.path.join(root_dir, "presto_amd", "host", "accelbatch_hip")
    a, b = tmp_path / "single", tmp_path / "batch"
    a.mkdir()
    b.mkdir()
    names = []
    for k, (N, ext) in enumerate([(1 << 19, ".fft"), (1 << 19, ".fft"), (1 << 19, ".dat"), (1 << 18, ".fft"), (1 << 18, ".fft")]):
        dt = 2.5e-4
        puls = [dict(f=77.7 + k, z=-11.0 + 4 * k, amp=0.3, duty=0.05), dict(f=201.0, z=3.0, amp=0.2, duty=0.1)]
        root = str(a / ("t%d_DM%.2f" % (k, 5.0 * k)))
        if ext == ".dat":
            synth.pulsar_series(N, dt, puls, seed=60 + k).astype(np.float32).tofile(root + ".dat")
            open(root + ".inf", "w").write(synth.inf_text(os.path.basename(root), N, dt))
        else:
            synth.write_fft(root, synth.make_fft(N, dt, puls, seed=60 + k), N, dt)
        names.append(os.path.basename(root) + ext)
    for n in os.listdir(str(a)):
        shutil.copy(str(a / n), str(b / n))
    flags = ["-zmax", "20", "-numharm", "8", "-sigma", "3.0", "-fhi", "1900"]
    for n in names:
        subprocess.check_output([one] + flags + [str(a / n)], timeout=170)
    out = subprocess.check_output([many] + flags + [str(b / n) for n in names], timeout=170).decode()
    assert "5 files in" in out
    made = sorted(n for n in os.listdir(str(a)) if "_ACCEL_" in n)
    assert len(made) == 3 * 4 + 2 and made == sorted(n for n in os.listdir(str(b)) if "_ACCEL_" in n)
    for n in made:
        assert open(str(a / n), "rb").read() == open(str(b / n), "rb").read(), n


def test_python_pipeline_equals_one_file_at_a_time(tmp_path):
    """pipeline.search_files (reader thread, two searches queued, each file's optimisation asked for after the
    next search has been collected) writes byte for byte what pipeline.search_one_file writes for each file on
    its own, a change of series length in mid-batch and a file without candidates included."""
    import shutil
    from presto_amd import api
    a, b = tmp_path / "single", tmp_path / "piped"
    a.mkdir()
    b.mkdir()
    names = []
    for k, N in enumerate([1 << 19, 1 << 19, 1 << 19, 1 << 18, 1 << 18, 1 << 19]):
        dt = 2.5e-4
        puls = [] if k == 1 else [dict(f=91.3 + k, z=7.0 - 3 * k, amp=0.3, duty=0.05), dict(f=155.0, z=-2.0, amp=0.2, duty=0.1)]
        root = str(a / ("p%d_DM%.2f" % (k, 3.0 * k)))
        synth.write_fft(root, synth.make_fft(N, dt, puls, seed=80 + k), N, dt)
        names.append(os.path.basename(root) + ".fft")
    for n in os.listdir(str(a)):
        shutil.copy(str(a / n), str(b / n))
    flags = ["-zmax", "20", "-numharm", "8", "-sigma", "6.0" , "-fhi", "1900", "-verbosity", "0"]
    gpu = api.AccelSearchGPU(0)
    opts = pipeline.parse_args(flags + [str(a / n) for n in names])
    alone = [pipeline.search_one_file(gpu, str(a / n), opts)[0] for n in names]
    gpu = api.AccelSearchGPU(0)
    files = [str(b / n) for n in names]
    piped = {i: cl for i, cl, _ in pipeline.search_files(gpu, files, list(range(len(files))), opts)}
    assert sorted(piped) == list(range(len(files)))
    made = sorted(n for n in os.listdir(str(a)) if "_ACCEL_" in n)
    assert made == sorted(n for n in os.listdir(str(b)) if "_ACCEL_" in n) and len(made) >= 3 * 4
    for n in made:
        assert open(str(a / n), "rb").read() == open(str(b / n), "rb").read(), n
    for i, cl in enumerate(alone):
        assert [(c.candnum, c.sigma, c.r, c.z) for c in cl] == [(c.candnum, c.sigma, c.r, c.z) for c in piped[i]]
