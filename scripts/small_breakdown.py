#!/usr/bin/env python3
"""Where a small trial's 0.2-0.3 ms go on the host: time inside each call of the pipelined loop."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch                                    # noqa: E402
import presto_amd as pa                         # noqa: E402
from presto_amd import synth                    # noqa: E402
from presto_amd.api import replay_candidates    # noqa: E402

N, dt, zmax, numharm = 1 << 20, 1e-4, 0, 8
dev = torch.device("cuda", 0)
spectra = []
for i in range(4):
    fft = synth.make_fft(N, dt, [dict(f=123.456789, z=0.0, amp=0.3, duty=0.03)], seed=i + 1)
    spectra.append(torch.from_numpy(fft.view(np.float32).copy()).to(dev))
g = pa.AccelSearchGPU(0)
setup = g.configure(N // 2, N, dt, zmax, numharm)
acc = {"load": 0.0, "launch": 0.0, "collect": 0.0, "replay": 0.0}


def launch(i):
    t = spectra[i % 4]
    t0 = time.perf_counter()
    g.load_fft_device(t.data_ptr(), N // 2, keepalive=t)
    t1 = time.perf_counter()
    g.run_launch()
    t2 = time.perf_counter()
    acc["load"] += t1 - t0
    acc["launch"] += t2 - t1


REPLAY_FIRST = len(sys.argv) > 1 and sys.argv[1] == "replay-first"
if len(sys.argv) > 1 and sys.argv[1] == "no-replay":
    replay_candidates = lambda setup, hits: (hits[:0], None, None)   # noqa: E731


def steps(k):
    launch(0)
    launch(1)
    for i in range(k):
        t0 = time.perf_counter()
        hits = g.run_collect(max_hits=1 << 16)
        t1 = time.perf_counter()
        acc["collect"] += t1 - t0
        if REPLAY_FIRST:
            replay_candidates(setup, hits)
            acc["replay"] += time.perf_counter() - t1
        if i + 2 < k:
            launch(i + 2)
        if not REPLAY_FIRST:
            t2 = time.perf_counter()
            replay_candidates(setup, hits)
            acc["replay"] += time.perf_counter() - t2


steps(20)
g.synchronize()
for k in acc:
    acc[k] = 0.0
t0 = time.perf_counter()
K = 400
steps(K)
g.synchronize()
el = time.perf_counter() - t0
print("%.0f searches/s, %.1f us per search; host time per search: " % (K / el, 1e6 * el / K)
      + ", ".join("%s %.1f us" % (k, 1e6 * v / K) for k, v in acc.items()))
g.close()

# the same replay with the GPU idle, on one trial's real hits
g2 = pa.AccelSearchGPU(0)
setup = g2.configure(N // 2, N, dt, zmax, numharm)
g2.load_fft_device(spectra[0].data_ptr(), N // 2, keepalive=spectra[0])
hits = g2.run()
t0 = time.perf_counter()
for _ in range(200):
    cands, sig, added = replay_candidates(setup, hits)
print("idle GPU: replay of %d real hits -> %d candidates: %.1f us" % (len(hits), len(cands), (time.perf_counter() - t0) / 200 * 1e6))
st = np.bincount(hits["stage"], minlength=4)
print("hits per stage", st, "power range", hits["power"].min(), hits["power"].max())
g2.close()
