"""What the coarse bound costs as its list grows: the benchmark's plane (2^23 points, zmax 200) with the
thresholds lowered by hand, 16 and 8 harmonics.  Per threshold scale: the bounded sum alone (hs_bound=3),
the coarse bound with no economic limit on its list (what a list of that length costs), with the default
limit (a fresh context: the walk is given up where it does not pay) and in steady state (a context that
remembers: its next trials go without).  Hits must be identical throughout.
python scripts/pf_dense_sweep.py [numharm ...]   (one GPU)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import presto_amd as pa                      # noqa: E402
from presto_amd import synth                 # noqa: E402
from presto_amd.api import plan              # noqa: E402

numbins, dt, zmax = 1 << 22, 64e-6, 200
SCALES = tuple(float(x) for x in os.environ.get("PF_SCALES", "1.0,0.9,0.88,0.87,0.85,0.8,0.76,0.72,0.68,0.6").split(","))
g = pa.AccelSearchGPU(0)
fft = synth.make_fft(2 * numbins, dt, dict(synth.CONFIGS["cfg2"])["pulsars"], seed=1000)
g.load_fft(fft)
g.set_tuning("hits_cap", 1 << 26)
LIMIT_PPM = int(os.environ.get("PF_LIMIT_PPM", g.get_tuning("pf_limit_ppm")))
print("pf_limit_ppm = %d" % LIMIT_PPM, flush=True)


def timed(reps):
    hits = g.run(max_hits=1 << 26)
    g.timings()
    for _ in range(reps):
        hits = g.run(max_hits=1 << 26)
    tm = g.timings()
    return hits, tm["correlate"]["total_ms"] / reps, tm["harmsum"]["total_ms"] / reps


for nh in [int(a) for a in sys.argv[1:]] or [16, 8]:
    print("numharm %d: scale | hits | listed %% | bounded sum alone (gen + sum) | list, no limit | default limit, fresh | "
          "default limit, remembered (16 trials) | worst / bounded" % nh, flush=True)
    for scale in SCALES:
        setup = plan(numbins, 2 * numbins, dt, zmax, nh, fhi=1e9)
        for s in range(setup.numharmstages):
            setup.powcut[s] *= scale
        g.create_subharminfos(setup.zmax)
        g.set_search(setup)
        g.set_profiling(True)
        groups = setup.numz * setup.num_search_blocks * 1868
        g.set_tuning("pf_list_cap", 0)
        g.set_tuning("hs_bound", 3)
        ref, cb, sb = timed(3)
        # the list without an economic limit, with room for a third of all (row, group)s
        g.set_tuning("hs_bound", 1)
        g.set_tuning("pf_memory", 0)
        g.set_tuning("pf_limit_ppm", 1000000)
        g.set_tuning("pf_list_cap", groups // 3)
        h1, c1, s1 = timed(3)
        listed = g.get_tuning("pf_listed")
        g.set_tuning("pf_list_cap", 0)
        # default limit, every trial tries (what a fresh context pays)
        g.set_tuning("pf_limit_ppm", LIMIT_PPM)
        h2, c2, s2 = timed(3)
        gave_up = g.get_tuning("pf_listed") == -1
        # ... and a context that remembers, over 16 trials
        g.set_tuning("pf_memory", 1)
        h3, c3, s3 = timed(16)
        same = ref.tobytes() == h1.tobytes() == h2.tobytes() == h3.tobytes()
        worst = max(c2 + s2, c3 + s3) / (cb + sb)
        print("%.2f | %8d | %s | %.2f + %.2f = %.2f | %.2f + %.2f = %.2f | %.2f + %.2f = %.2f%s | %.2f + %.2f = %.2f | %.3f%s" % (
            scale, len(ref), "%.3f" % (100.0 * listed / groups) if listed >= 0 else "> 33", cb, sb, cb + sb, c1, s1, c1 + s1,
            c2, s2, c2 + s2, " (given up)" if gave_up else "", c3, s3, c3 + s3, worst, "" if same else "  HITS DIFFER"), flush=True)
g.close()
