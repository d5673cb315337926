#!/bin/bash
# PMC passes for the Welch kernel (separate runs, no tracing domains mixed in).
# usage (on the GPU box): bash tools/pmc_welch.sh <out-subdir> [nperseg]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
NP=${2:-4096}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
run() { # name, counters
  rocprofv3 --pmc $2 --output-format csv -d $OUT/$1 -- python3 $GRAFT_REPO_ROOT/tools/run_kernel.py welch --reps 2 --nperseg $NP > $OUT/$1.log 2>&1
}
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
run b "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES"
run c "FETCH_SIZE"
run d "WRITE_SIZE"
# FETCH_SIZE calibration for K2's 2-byte-per-lane pattern (tools/calib_fetch.hip, built in-tree)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/calib -- $GRAFT_REPO_ROOT/tools/calib_fetch > $OUT/calib.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/run_kernel.py welch --reps 5 --nperseg $NP > $OUT/trace.log 2>&1
find $OUT -name "*.csv" | head -30
