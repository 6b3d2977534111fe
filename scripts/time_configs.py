"""Whole-search time and kernel split of the large configurations on one GPU."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import presto_amd as pa                      # noqa: E402
from presto_amd import synth                 # noqa: E402

SUMF2 = {1: 0.0, 2: 0.25, 4: 0.875, 8: 2.1875, 16: 4.84375, 32: 10.171875}
CASES = {"cfg1": (1 << 19, 1e-4, 0, 8), "cfg2": (1 << 22, 64e-6, 200, 16),
         "cfg5": (1 << 23, 64e-6, 1200, 32), "cfg4": (1 << 26, 64e-6, 300, 16)}
g = pa.AccelSearchGPU(0)
for name in sys.argv[1:] or ["cfg1", "cfg2", "cfg5", "cfg4"]:
    numbins, dt, zmax, nh = CASES[name]
    fft = synth.noise_fft(numbins, 5)
    setup = g.configure(numbins, 2 * numbins, dt, zmax, nh, fhi=1e9)
    g.load_fft(fft)
    g.set_profiling(True)
    hits = g.run(max_hits=1 << 22)
    g.timings()
    t0 = time.perf_counter()
    hits = g.run(max_hits=1 << 22)
    wall = time.perf_counter() - t0
    tm = g.timings()
    alg = 4.0 * (1 + SUMF2[nh]) * setup.numz * 7470 * setup.num_search_blocks
    hs = tm["harmsum"]["total_ms"]
    print("%s: 2^%d bins zmax %d numharm %d: plane %.1f GB, %d blocks; search %.1f ms wall; prep %.2f correlate %.3f "
          "harmsum %.1f ms = %.0f GB/s (%.1f%% of 8 TB/s); %d hits" % (
              name, int(np.log2(numbins)), zmax, nh, setup.plane_bytes / 1e9, setup.num_search_blocks, 1e3 * wall,
              tm["block_prep"]["total_ms"], tm["correlate"]["total_ms"], hs, alg / hs / 1e6, alg / hs / 1e6 / 80,
              len(hits)), flush=True)
g.close()
