#!/bin/bash
# bash scripts/pf_walk_prof.sh a.so b.so ...: kernel times of scripts/pf_walk_time.py under rocprofv3 with each library build
R=${GRAFT_REPO_ROOT:-$(pwd)}
cp $R/presto_amd/libaccelgpu.so /tmp/libaccelgpu_cur.so
cd /tmp && export TMPDIR=/tmp
for so in "$@"; do
  cp $R/$so $R/presto_amd/libaccelgpu.so
  rm -rf /tmp/pfw
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pfw -o t -- python3 $R/scripts/pf_walk_time.py ${NH:-16} > /tmp/pfw.log 2>&1 || { tail -5 /tmp/pfw.log; break; }
  echo "== $so"; grep numharm /tmp/pfw.log
  python3 - <<'PY'
import csv
for r in csv.DictReader(open("/tmp/pfw/t_kernel_stats.csv")):
    if any(k in r["Name"] for k in ("prefilter", "flagged", "correlate_rr", "group_fix")):
        print("   %-60s %4s calls  min %.3f ms" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["MinNs"]) / 1e6))
PY
done
cp /tmp/libaccelgpu_cur.so $R/presto_amd/libaccelgpu.so
