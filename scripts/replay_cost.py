#!/usr/bin/env python3
"""Cost of the host replay of one small trial's hits: the C call alone and the Python wrapper around it."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import presto_amd.api as api

s = api.plan(1 << 19, 1 << 20, 1e-4, 0, 8)
rng = np.random.default_rng(0)
n = 624
hits = np.zeros(n, dtype=api.HIT_DTYPE)
hits["power"] = rng.uniform(17, 40, n)
hits["stage"] = rng.integers(0, 4, n)
hits["rind"] = np.sort(rng.integers(300, 1000, n)) + 20000
L = api.load_library()
cands = np.zeros(n, dtype=api.CAND_DTYPE)
sig = np.zeros(n, dtype=np.float32)
added = np.zeros(n, dtype=np.int32)
nc = C.c_int64(0)
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(500):
        L.accelgpu_replay_candidates(C.byref(s), api._ptr(hits), n, api._ptr(cands), n, C.byref(nc), api._ptr(sig), api._ptr(added))
    t1 = time.perf_counter()
    for i in range(500):
        api.replay_candidates(s, hits)
    t2 = time.perf_counter()
    for i in range(500):
        for h in hits[:8]:
            L.accelgpu_candidate_sigma(float(h["power"]), 2, 1e5)
    t3 = time.perf_counter()
    print("C call %.1f us, wrapper %.1f us, 8 scalar sigma calls %.1f us" % ((t1 - t0) / 500 * 1e6, (t2 - t1) / 500 * 1e6, (t3 - t2) / 500 * 1e6), flush=True)
