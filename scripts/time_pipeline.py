"""End-to-end rate of the batch driver on one GPU: DM-trial .fft files on local disk ->
search -> optimisation -> ACCEL/.cand/.txtcand files -> sifting."""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from presto_amd import pipeline, synth       # noqa: E402

nfiles = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ndistinct = min(nfiles, 8)
cfg = dict(synth.CONFIGS["cfg2"])
tmp = tempfile.mkdtemp(prefix="pipe_", dir="/tmp")
files = []
roots = []
for k in range(nfiles):
    root = os.path.join(tmp, "beam_DM%.2f" % (10.0 + k))
    if k < ndistinct:
        synth.write_fft(root, synth.make_fft(cfg["N"], cfg["dt"], cfg["pulsars"], seed=1000 + k), cfg["N"], cfg["dt"])
    else:                                    # further trials: other names for the same bytes (results go by the name)
        os.symlink(roots[k % ndistinct] + ".fft", root + ".fft")
        os.symlink(roots[k % ndistinct] + ".inf", root + ".inf")
    roots.append(root)
    files.append(root + ".fft")
opts = pipeline.parse_args(["-zmax", "200", "-numharm", "16", "-fhi", "1e9", "-verbosity", "0",
                            "-o", os.path.join(tmp, "beam")] + files)
for rep in range(2):
    t0 = time.perf_counter()
    cl, secs = pipeline.run(sorted(files), opts)
    dt = time.perf_counter() - t0
    print("pass %d: %d files in %.3f s = %.1f files/s (%.1f files/s after the first file: context, template bank and "
          "plane are set up once per process); %d sifted candidates" % (
              rep, nfiles, dt, nfiles / dt, (nfiles - 1) / max(sum(secs[1:]), 1e-9), len(cl)))
    print("    ms per file: " + "  ".join("%s %.2f" % (k, 1e3 * v / nfiles) for k, v in pipeline.TIMES.items()))
    pipeline.TIMES.clear()

# the native program on the same files
import subprocess
exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "presto_amd", "host", "accelbatch_hip")
for rep in range(2):
    out = subprocess.check_output([exe, "-zmax", "200", "-numharm", "16", "-fhi", "1e9", "-quiet"] + sorted(files)).decode()
    print("accelbatch_hip pass %d: %s" % (rep, out.strip().splitlines()[-1]))
