"""FETCH_SIZE calibration: harmonic-sum kernel with 1, 2 and 16 harmonics on a
cfg2-sized plane.  numharm=1 streams the plane exactly once (6.74 GB)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import presto_amd as pa
from presto_amd import synth
numbins = 1 << 22
fft = synth.noise_fft(numbins, 3)
g = pa.AccelSearchGPU(0)
for nh in (1, 2, 16):
    setup = g.configure(numbins, 2 * numbins, 64e-6, 200, nh, fhi=1e9)
    g.load_fft(fft)
    g.fill_plane()
    for _ in range(3):
        h = g.search_plane()
    print(nh, "cells*4 GB", setup.numz * 7470 * setup.num_search_blocks * 4 / 1e9, "hits", len(h))
