#!/bin/bash
# bash scripts/lib_times.sh a.so b.so ...: time_configs.py cfg2 with each library build in place (same box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cp presto_amd/libaccelgpu.so /tmp/libaccelgpu_cur.so
for so in "$@" /tmp/libaccelgpu_cur.so; do
  cp $so presto_amd/libaccelgpu.so
  echo "== $so"
  timeout -k 10 200 python3 scripts/time_configs.py ${CFGS:-cfg2 cfg2} 2>&1 | tail -2 | cut -c1-200 || exit 1
done
cp /tmp/libaccelgpu_cur.so presto_amd/libaccelgpu.so
