#!/bin/bash
# The search with candidate optimisation beside it, against the optimisation's threads per candidate (same box):
#   bash scripts/opt_nt_ab.sh 128 256 512 1024
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/optnt
for nt in "$@"; do
  env ACCELGPU_OPT_NT=$nt timeout -k 10 200 python3 bench.py --no-cpu > gpurun_out/optnt/$nt.json 2> gpurun_out/optnt/$nt.err || { echo "failed at $nt"; break; }
  python3 - gpurun_out/optnt/$nt.json $nt <<'PY'
import json, sys
d = json.loads([x for x in open(sys.argv[1]) if x.startswith('{')][-1])
print("opt_nt %4s: %.1f searches/s alone, %.1f with optimisation (%.2f ms per trial's %.1f candidates), %.1f from pinned uploads" % (
    sys.argv[2], d["value"], d["with_optimization_searches_per_s"], d["optimize_ms_per_trial"], d["cands_optimized_per_trial"], d["pcie_inclusive_searches_per_s"]))
PY
done
