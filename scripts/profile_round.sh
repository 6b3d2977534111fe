#!/bin/bash
# Usage (on the GPU box): bash scripts/profile_round.sh rNN
# Writes rocprofv3 kernel-trace stats and separate PMC passes for bench.py into gpurun_out/<tag>/.
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/bench.py --steps 20 --warmup 3 --timed-only > $OUT/bench_under_trace.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 $R/bench.py --steps 3 --warmup 1 --timed-only > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 $R/bench.py --steps 3 --warmup 1 --timed-only > $OUT/pmc_write.log 2>&1
cd $R
python3 scripts/pmc_sum.py $OUT/pmc_fetch/f_counter_collection.csv accelk > $OUT/pmc_summary.txt
python3 scripts/pmc_sum.py $OUT/pmc_write/w_counter_collection.csv accelk >> $OUT/pmc_summary.txt
python3 scripts/pmc_json.py $OUT/pmc_fetch/f_counter_collection.csv $OUT/pmc_write/w_counter_collection.csv $OUT/pmc_kernels.json
cat $OUT/trace/t_kernel_stats.csv | cut -c1-220
cat $OUT/pmc_summary.txt
grep '"metric"' $OUT/bench_under_trace.log | cut -c1-600
