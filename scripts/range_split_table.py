"""One long series over several GPUs: what each rank generates and sums, for equal searched ranges and for
ranges of equal predicted work, and the predicted time against one GPU.  Host arithmetic only (no GPU).
python scripts/range_split_table.py [log2N zmax numharm]   (default: 27 300 16, the 2^27-point configuration)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from presto_amd import batch                               # noqa: E402
from presto_amd.api import plan, block_range_need          # noqa: E402

log2n, zmax, nh = [int(a) for a in sys.argv[1:4]] if len(sys.argv) >= 4 else (27, 300, 16)
N = 1 << log2n
s = plan(N // 2, N, 64e-6, zmax, nh, fhi=1e9)
nb, npre = s.num_search_blocks, s.num_prepop_blocks
full_rows = (nb + npre) * s.numz
one = batch.GEN_COST * full_rows + batch.SUM_COST * nb * s.numz
print("2^%d points, zmax %d, numharm %d: %d search + %d pre-population blocks x %d rows; one GPU: %.1f ms predicted "
      "(%.1f generation + %.1f sum)" % (log2n, zmax, nh, nb, npre, s.numz, one, batch.GEN_COST * full_rows,
                                        batch.SUM_COST * nb * s.numz))
for world in (2, 4, 8):
    for name, ranges in (("equal searched ranges", batch.split_blocks(nb, world)),
                         ("equal predicted work", batch.split_blocks_by_cost(s, world))):
        print("world %d, %s:" % (world, name))
        costs = []
        for r, (a, b) in enumerate(ranges):
            blocks, rows = block_range_need(s, a, b - a)
            c = batch.range_cost(s, a, b - a)
            costs.append(c)
            print("  rank %d: searches blocks [%5d, %5d) = %5.1f %%, generates %5d blocks, %9d rows = %5.1f %% of the plane; %.1f ms"
                  % (r, a, b, 100.0 * (b - a) / nb, blocks, rows, 100.0 * rows / full_rows, c))
        print("  slowest rank %.1f ms, mean %.1f ms (max / mean %.2f): %.2f x one GPU" % (
            max(costs), sum(costs) / world, max(costs) / (sum(costs) / world), one / max(costs)))
