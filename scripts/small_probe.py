#!/usr/bin/env python3
"""Is host code slower while GPU work is in flight?  Times the bare C replay call and a fixed
pure-CPU calibration loop inside the pipelined small-trial loop and again with the GPU idle."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch                                    # noqa: E402
import presto_amd as pa                         # noqa: E402
import presto_amd.api as api                    # noqa: E402
from presto_amd import synth                    # noqa: E402

N, dt, zmax, numharm = 1 << 20, 1e-4, 0, 8
dev = torch.device("cuda", 0)
spectra = []
for i in range(4):
    fft = synth.make_fft(N, dt, [dict(f=123.456789, z=0.0, amp=0.3, duty=0.03)], seed=i + 1)
    spectra.append(torch.from_numpy(fft.view(np.float32).copy()).to(dev))
g = pa.AccelSearchGPU(0)
setup = g.configure(N // 2, N, dt, zmax, numharm)
L = api.load_library()
cap = 1 << 16
cands = np.zeros(cap, dtype=api.CAND_DTYPE)
sig = np.zeros(cap, dtype=np.float32)
added = np.zeros(cap, dtype=np.int32)
nc = C.c_int64(0)
acc = {"replay_c": 0.0, "calib": 0.0}


def calib():
    x = 0
    for i in range(300):
        x += i * i
    return x


def launch(i):
    t = spectra[i % 4]
    g.load_fft_device(t.data_ptr(), N // 2, keepalive=t)
    g.run_launch()


def steps(k, busy):
    if busy:
        launch(0)
        launch(1)
    hits = None
    for i in range(k):
        if busy:
            hits = g.run_collect(max_hits=cap)
            if i + 2 < k:
                launch(i + 2)
        t0 = time.perf_counter()
        L.accelgpu_replay_candidates(C.byref(setup), api._ptr(hits), len(hits), api._ptr(cands), cap, C.byref(nc),
                                     api._ptr(sig), api._ptr(added))
        t1 = time.perf_counter()
        calib()
        t2 = time.perf_counter()
        acc["replay_c"] += t1 - t0
        acc["calib"] += t2 - t1
    return hits


hits = steps(20, True)
g.synchronize()
for mode in (True, False, True, False):
    for k in acc:
        acc[k] = 0.0
    K = 300
    if mode:
        hits = steps(K, True)
    else:
        saved = hits
        steps_hits = saved

        def idle(k):
            for i in range(k):
                t0 = time.perf_counter()
                L.accelgpu_replay_candidates(C.byref(setup), api._ptr(saved), len(saved), api._ptr(cands), cap,
                                             C.byref(nc), api._ptr(sig), api._ptr(added))
                t1 = time.perf_counter()
                calib()
                t2 = time.perf_counter()
                acc["replay_c"] += t1 - t0
                acc["calib"] += t2 - t1
        idle(K)
    g.synchronize()
    print("%s: C replay %.1f us, calibration loop %.1f us (%d hits)" % ("GPU busy" if mode else "GPU idle",
          1e6 * acc["replay_c"] / K, 1e6 * acc["calib"] / K, len(hits)), flush=True)
g.close()
