#!/bin/bash
# Usage (on the GPU box): bash scripts/profile_calib.sh rNN
# FETCH_SIZE / WRITE_SIZE of the harmonic-sum kernels with 1, 2 and 16 harmonics on a cfg2-sized plane
# (scripts/calib.py): one harmonic streams the plane (or its 16-bit bounds) exactly once.
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -o f -- python3 $R/scripts/calib.py > $OUT/calib_fetch.log 2>&1
cd $R
python3 scripts/pmc_sum.py $OUT/calib_fetch/f_counter_collection.csv k_harmsum > $OUT/calib_summary.txt
cat $OUT/calib_fetch.log | tail -4
cat $OUT/calib_summary.txt
