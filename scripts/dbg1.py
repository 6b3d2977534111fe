import sys, numpy as np
sys.path.insert(0, '.')
import presto_amd as pa
from presto_amd import synth
from oracle import binding as ob
fft, c = synth.config_fft("cfg1", N=1 << 19)
g = pa.AccelSearchGPU(0)
setup = g.configure(fft.size, c["N"], c["dt"], 0, 8, fhi=1e9)
g.load_fft(fft)
s = ob.Search(fft, c["N"], c["dt"], 0, 8, fhi=1e9)
g.fill_plane()
P = g.plane()
s.set_plane(P)
blk = 1
rlo_b = int(s.rlo) + blk * 3735
pw = np.ascontiguousarray(s.plane()[:, 2 * rlo_b: 2 * rlo_b + 7470])
for stage in range(1, 3):
    for h in range(1, 1 << stage, 2):
        s.inmem_add(pw, rlo_b, 1 << stage, h)
    got = g.summed_block(blk, stage)
    bad = np.nonzero(got.view(np.uint32) != pw.view(np.uint32))
    print("stage", stage, "nbad", len(bad[0]), "first", bad[1][:5] if len(bad[0]) else None)
    if len(bad[0]):
        j = bad[1][0]
        rr = 2 * rlo_b + j
        print("j", j, "rr", rr, "got", got[0, j], "want", pw[0, j])
        comp = [P[0, rr]] + [P[0, int(rr * h / (1 << st) + 0.5)] for st in range(1, stage + 1) for h in range(1, 1 << st, 2)]
        print("components", comp, "sum", np.sum(np.array(comp, dtype=np.float32)))
        print("rlen", setup.rlen, "ncols total", setup.num_search_blocks * 7470, "col0", 2 * int(s.rlo))
