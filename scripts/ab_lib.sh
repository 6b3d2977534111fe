#!/bin/bash
# Same-box A/B of two builds of the library: bash scripts/ab_lib.sh <old.so> [bench args]
# Runs bench.py --timed-only with the old library in place, then with the current one.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
OLD=$1; shift
mkdir -p gpurun_out/ab
cp presto_amd/libaccelgpu.so /tmp/libaccelgpu_new.so
show() {
python3 - "$1" "$2" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith('{')][-1]
d = json.loads(l)
print(sys.argv[2], "%.2f searches/s  %.3f ms/step" % (d["value"], d["ms_per_step"]), {k: round(v, 3) for k, v in d["kernels_ms_per_step"].items()}, "hits", d["hits_per_trial"])
PY
}
for rep in 1 2; do
  cp $OLD presto_amd/libaccelgpu.so
  timeout -k 10 150 python3 bench.py --steps 30 --warmup 4 --timed-only "$@" > gpurun_out/ab/old$rep.json 2> gpurun_out/ab/old$rep.err
  show gpurun_out/ab/old$rep.json old$rep
  cp /tmp/libaccelgpu_new.so presto_amd/libaccelgpu.so
  timeout -k 10 150 python3 bench.py --steps 30 --warmup 4 --timed-only "$@" > gpurun_out/ab/new$rep.json 2> gpurun_out/ab/new$rep.err
  show gpurun_out/ab/new$rep.json new$rep
done
