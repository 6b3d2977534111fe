#!/bin/bash
# bash scripts/lib_std_times.sh old.so: standard-mode and spectrum timings with the old and the current library (same box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cp presto_amd/libaccelgpu.so /tmp/libaccelgpu_cur.so
for so in "$@" /tmp/libaccelgpu_cur.so; do
  cp $so presto_amd/libaccelgpu.so
  echo "== $so"
  timeout -k 10 200 python3 scripts/time_standard.py 2>&1 | tail -2 || exit 1
  timeout -k 10 200 python3 scripts/time_spectrum.py 2>&1 | tail -4 || exit 1
done
cp /tmp/libaccelgpu_cur.so presto_amd/libaccelgpu.so
