#!/bin/bash
# PMC passes over the coarse bound's kernel (k_prefilter_*) in bench.py --timed-only: issue, wait and load-path activity.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pf_pmc
mkdir -p $OUT
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p1 -o a -- python3 $R/bench.py --steps 3 --warmup 1 --timed-only > $OUT/p1.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -o b -- python3 $R/bench.py --steps 3 --warmup 1 --timed-only > $OUT/p2.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc TA_BUSY_avr TA_BUSY_max TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/p3 -o c -- python3 $R/bench.py --steps 3 --warmup 1 --timed-only > $OUT/p3.log 2>&1
cd $R
for f in $OUT/p1/a_counter_collection.csv $OUT/p2/b_counter_collection.csv $OUT/p3/c_counter_collection.csv; do
  [ -f $f ] && python3 scripts/pmc_sum.py $f k_prefilter
  [ -f $f ] && python3 scripts/pmc_sum.py $f k_correlate
done
exit 0
