#!/bin/bash
# bash scripts/lib_cmd.sh "<command>" a.so b.so ...: runs the command with each library build in place (same box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
CMD=$1; shift
cp presto_amd/libaccelgpu.so /tmp/libaccelgpu_cur.so
for so in "$@"; do
  cp $so presto_amd/libaccelgpu.so
  echo "== $so"
  timeout -k 10 200 $CMD 2>&1 | tail -${TAIL:-6} | cut -c1-200 || break
done
cp /tmp/libaccelgpu_cur.so presto_amd/libaccelgpu.so
