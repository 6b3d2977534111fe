#!/bin/bash
# Same-box comparison of environment knobs of the current library:
#   bash scripts/env_ab.sh "A=1" "B=2 C=3" ...   (each argument one configuration; "-" = defaults)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/envab
i=0
for cfg in "$@"; do
  i=$((i+1))
  if [ "$cfg" = "-" ]; then envs=""; else envs="$cfg"; fi
  env $envs timeout -k 10 150 python3 bench.py --steps 30 --warmup 4 --timed-only > gpurun_out/envab/$i.json 2> gpurun_out/envab/$i.err
  python3 - gpurun_out/envab/$i.json "$cfg" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith('{')][-1]
d = json.loads(l)
print(sys.argv[2], "%.2f searches/s  %.3f ms/step" % (d["value"], d["ms_per_step"]), {k: round(v, 3) for k, v in d["kernels_ms_per_step"].items()}, "hits", d["hits_per_trial"])
PY
done
