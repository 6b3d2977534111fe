"""Times candidate optimisation: GPU (one workgroup per candidate) vs the CPU oracle.
Lives under tests/ because it runs the oracle (test infrastructure)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import presto_amd as pa
from presto_amd import synth
from oracle import binding as ob
N = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 21)
fft, c = synth.config_fft("cfg2", N=N)
g = pa.AccelSearchGPU(0)
g.configure(fft.size, c["N"], c["dt"], c["zmax"], c["numharm"], fhi=1e9)
g.load_fft(fft)
hits, cands, sig, added = g.search()
kept = pa.api.sort_and_eliminate(cands)
print("N=%d: %d hits, %d candidates, %d after harmonic elimination; numharm counts %s" % (
    N, len(hits), len(cands), len(kept), np.bincount(kept["numharm"])[[1, 2, 4, 8, 16]]))
g.optimize(kept[:2])
for polish in (True, False):
    t = time.perf_counter(); out = g.optimize(kept, polish=polish); tg = time.perf_counter() - t
    print("GPU optimize polish=%d: %.1f ms for %d candidates" % (polish, tg * 1e3, len(kept)))
s = ob.Search(fft, c["N"], c["dt"], c["zmax"], c["numharm"], fhi=1e9, alloc_plane=False)
oc = np.zeros(len(kept), dtype=ob.CAND_DTYPE)
for k in ("power", "sigma", "numharm", "r", "z"):
    oc[k] = kept[k]
nsub = min(len(kept), 12)
t = time.perf_counter(); o = s.optimize(oc[:nsub], polish=True, eliminate=False); tc = time.perf_counter() - t
print("CPU oracle polish=1: %.1f ms for %d candidates (1 core) -> %.1f ms for all" % (tc * 1e3, nsub, tc * 1e3 * len(kept) / nsub))
out = g.optimize(kept[:nsub], polish=True)
print("max |dr| %.2e  max |dz| %.2e" % (np.abs(o["hirs"][:, 0] - out["hirs"][:, 0]).max(), np.abs(o["hizs"][:, 0] - out["hizs"][:, 0]).max()))
