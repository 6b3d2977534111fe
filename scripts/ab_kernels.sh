#!/bin/bash
# Per-kernel times of several builds of the library on one box: bash scripts/ab_kernels.sh a.so b.so ...
# (rocprofv3 --kernel-trace --stats of bench.py --timed-only with each library in place)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
cp presto_amd/libaccelgpu.so /tmp/libaccelgpu_cur.so
for so in "$@"; do
  cp $so presto_amd/libaccelgpu.so
  n=$(basename $so .so)
  rm -rf gpurun_out/abk_$n
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abk_$n -o p -- python3 bench.py --steps 20 --warmup 3 --timed-only > gpurun_out/abk_$n.json 2> gpurun_out/abk_$n.err || { echo "failed: $so"; break; }
  echo "== $so"
  f=$(find gpurun_out/abk_$n -name '*kernel_stats.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["TotalDurationNs"]) > 2e5:
        print("  %-90s calls %4s avg %9.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
cp /tmp/libaccelgpu_cur.so presto_amd/libaccelgpu.so
