"""One long series over several GPUs, rehearsed on one: every rank's block range of a split (equal predicted
work) is searched in turn on the same GPU -- plane generation of the rows that rank needs, harmonic sum of its
columns -- and timed; the merged hits must be those of the unsplit search.  What `world` GPUs would take is the
slowest rank's time (the ranks share nothing but the final gather of hits).
python scripts/range_split_measure.py [cfg4|cfg2]   (one GPU; cfg4 holds a 161.6 GB plane)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import presto_amd as pa                      # noqa: E402
from presto_amd import batch, synth          # noqa: E402

CASES = {"cfg2": (1 << 22, 64e-6, 200, 16), "cfg4": (1 << 26, 64e-6, 300, 16)}
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
numbins, dt, zmax, nh = CASES[name]
g = pa.AccelSearchGPU(0)
fft = synth.noise_fft(numbins, 5, lines=[(0.31 * numbins + 0.5, 37.0, 14.0), (0.87 * numbins, -150.0, 13.0)])
setup = g.configure(numbins, 2 * numbins, dt, zmax, nh, fhi=1e9)
g.load_fft(fft)
g.set_profiling(True)


def timed():
    """One untimed search, then three timed ones: the median by total (single runs showed one-off outliers of 5 ms)."""
    g.run(max_hits=1 << 22)
    runs = []
    for _ in range(3):
        g.timings()
        hits = g.run(max_hits=1 << 22)
        tm = g.timings()
        runs.append((tm["block_prep"]["total_ms"] + tm["correlate"]["total_ms"], tm["harmsum"]["total_ms"]))
    gen, hs = sorted(runs, key=sum)[1]
    return hits, gen, hs


whole, gen1, sum1 = timed()
print("%s: %d search blocks x %d rows, plane %.1f GB; one GPU: generation %.1f + sum %.1f = %.1f ms, %d hits" % (
    name, setup.num_search_blocks, setup.numz, setup.plane_bytes / 1e9, gen1, sum1, gen1 + sum1, len(whole)), flush=True)
for world in (2, 4, 8):
    ranges = batch.split_blocks_by_cost(setup, world)
    parts, times = [], []
    for rank, (a, b) in enumerate(ranges):
        nblocks = g.set_block_range(a, b - a)
        hits, gen, hs = timed()
        parts.append(hits)
        times.append(gen + hs)
        print("  world %d rank %d: blocks [%5d, %5d), generates %5d blocks: generation %.1f + sum %.1f = %.1f ms (predicted %.1f)" % (
            world, rank, a, b, nblocks, gen, hs, gen + hs, batch.range_cost(setup, a, b - a)), flush=True)
    merged = batch.reference_order(np.concatenate(parts))
    print("world %d: slowest rank %.1f ms, mean %.1f (max / mean %.2f): %.2f x one GPU; merged hits identical: %s" % (
        world, max(times), np.mean(times), max(times) / np.mean(times), (gen1 + sum1) / max(times),
        np.array_equal(merged, whole)), flush=True)
    g.set_block_range(0, setup.num_search_blocks)
g.close()
