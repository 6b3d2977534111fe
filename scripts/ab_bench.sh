#!/bin/bash
# bash scripts/ab_bench.sh a.so b.so ...: bench.py --timed-only with each library build in place, twice round (same box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cp presto_amd/libaccelgpu.so /tmp/libaccelgpu_cur.so
for round in 1 2; do
  for so in "$@"; do
    cp $so presto_amd/libaccelgpu.so
    timeout -k 10 200 python bench.py --timed-only --steps 40 --warmup 5 > /tmp/ab_bench.json 2>/tmp/ab_bench.err || { tail -5 /tmp/ab_bench.err; break 2; }
    python - "$so" <<'PY'
import json, sys
b = json.load(open('/tmp/ab_bench.json'))
k = b["kernels_ms_per_step"]
print("%-28s %.1f searches/s  step %.3f ms  correlate %.3f  harmsum %.3f  hits %s" % (sys.argv[1], b["value"], b["ms_per_step"], k["correlate"], k["harmsum"], b.get("hits_per_trial")))
PY
  done
done
cp /tmp/libaccelgpu_cur.so presto_amd/libaccelgpu.so
