"""Time of the coarse bound's walk with the list forced (hs_bound = 4, no sample): python scripts/pf_walk_time.py [numharm]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import presto_amd as pa                      # noqa: E402
from presto_amd import synth                 # noqa: E402

nh = int(sys.argv[1]) if len(sys.argv) > 1 else 16
numbins, dt, zmax = 1 << 22, 64e-6, 200
g = pa.AccelSearchGPU(0)
fft = synth.make_fft(2 * numbins, dt, dict(synth.CONFIGS["cfg2"])["pulsars"], seed=1000)
setup = g.configure(numbins, 2 * numbins, dt, zmax, nh, fhi=1e9)
g.load_fft(fft)
g.set_profiling(True)
g.set_tuning("hs_bound", 4)
g.set_tuning("pf_sample", 0)
for groups, nt in ((1, 256), (1, 512), (1, 1024), (2, 256), (2, 512), (2, 1024)):
    g.set_tuning("pf_groups", groups)
    try:
        g.set_tuning("pf_nt", nt)
    except Exception:
        if nt != 256:
            continue
    hits = g.run(max_hits=1 << 23)
    g.timings()
    for _ in range(3):
        hits = g.run(max_hits=1 << 23)
    tm = g.timings()
    print("numharm %d pf_groups %d nt %d: correlate %.3f harmsum %.3f ms, %d hits, listed %d" % (
        nh, groups, nt, tm["correlate"]["total_ms"] / 3, tm["harmsum"]["total_ms"] / 3, len(hits), g.get_tuning("pf_listed")), flush=True)
g.close()
