#!/bin/bash
# PMC passes over the plane-generation kernels (correlate): VALU / LDS activity and cycles.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/corr_pmc
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY --output-format csv -d $OUT/p1 -o a -- python3 $R/scripts/corr_sweep.py > $OUT/p1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -o b -- python3 $R/scripts/corr_sweep.py > $OUT/p2.log 2>&1
cd $R
python3 scripts/pmc_sum.py $OUT/p1/a_counter_collection.csv k_correlate
python3 scripts/pmc_sum.py $OUT/p2/b_counter_collection.csv k_correlate
tail -1 $OUT/p1.log
