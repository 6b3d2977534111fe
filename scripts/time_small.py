"""Pipelined search rate for small trials (launch- and host-bound regime): cfg1 = 2^20 points, zmax 0, numharm 8."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch                                    # noqa: E402
import presto_amd as pa                         # noqa: E402
from presto_amd import synth                    # noqa: E402
from presto_amd.api import replay_candidates    # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
zmax = int(sys.argv[2]) if len(sys.argv) > 2 else 0
numharm = int(sys.argv[3]) if len(sys.argv) > 3 else 8
N, dt = 1 << log2n, 1e-4
dev = torch.device("cuda", 0)
spectra = []
for i in range(4):
    fft = synth.make_fft(N, dt, [dict(f=123.456789, z=0.0, amp=0.3, duty=0.03)], seed=i + 1)
    spectra.append(torch.from_numpy(fft.view(np.float32).copy()).to(dev))
g = pa.AccelSearchGPU(0)
setup = g.configure(N // 2, N, dt, zmax, numharm)
def launch(i):
    t = spectra[i % 4]
    g.load_fft_device(t.data_ptr(), N // 2, keepalive=t)
    g.run_launch()
def steps(k):
    launch(0); launch(1)
    nh = 0
    for i in range(k):
        hits = g.run_collect(max_hits=1 << 16)
        if i + 2 < k:
            launch(i + 2)
        cands, _, _ = replay_candidates(setup, hits)
        nh += len(hits)
    return nh / k
steps(20)
g.synchronize()
# several rounds: the first second or so after start-up runs the host code several times slower
# (seen on every box: the C replay of one trial's hits takes 160 us at first and 40 us later)
k = 400
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 6
for rnd in range(rounds):
    t0 = time.perf_counter()
    nh = steps(k)
    g.synchronize()
    el = time.perf_counter() - t0
    print("2^%d points, zmax %d, numharm %d, round %d: %.0f searches/s (%.3f ms per search), %.0f hits per trial" % (
        log2n, zmax, numharm, rnd, k / el, 1e3 * el / k, nh), flush=True)
