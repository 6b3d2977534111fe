#!/bin/bash
# The search with candidate optimisation beside it, with each of several builds of the library in place (same box):
#   bash scripts/opt_lib_ab.sh a.so b.so ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/optlib
cp presto_amd/libaccelgpu.so /tmp/libaccelgpu_cur.so
for so in "$@"; do
  cp $so presto_amd/libaccelgpu.so
  n=$(basename $so .so)
  timeout -k 10 200 python3 bench.py --no-cpu > gpurun_out/optlib/$n.json 2> gpurun_out/optlib/$n.err || { echo "failed at $so"; break; }
  python3 - gpurun_out/optlib/$n.json $n <<'PY'
import json, sys
d = json.loads([x for x in open(sys.argv[1]) if x.startswith('{')][-1])
print("%s: %.1f searches/s alone, %.1f with optimisation (%.2f ms per trial's %.1f candidates alone)" % (
    sys.argv[2], d["value"], d["with_optimization_searches_per_s"], d["optimize_ms_per_trial"], d["cands_optimized_per_trial"]))
PY
done
cp /tmp/libaccelgpu_cur.so presto_amd/libaccelgpu.so
