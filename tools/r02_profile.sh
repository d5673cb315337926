#!/bin/bash
# Round-2 evidence run (on the GPU box, from the repo root):  bash tools/r02_profile.sh <tag>
#   1. bench.py plain (the JSON line)                         -> gpurun_out/<tag>/bench.json
#   2. rocprofv3 --kernel-trace --stats of the same command   -> gpurun_out/<tag>/bench_trace/
#   3. settled-clock SOLO traces of the reported kernels      -> gpurun_out/<tag>/solo_<kernel>/
#   4. PMC passes of K2 (separate runs, counters only)        -> gpurun_out/<tag>/pmc/
#   5. micro-benchmark MFMA pass 0                            -> gpurun_out/<tag>/ubench_mfma_pass0.txt
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
$ROOT/tools/ubench_mfma_pass0 > $OUT/ubench_mfma_pass0.txt 2>&1; echo "ubench rc=$?"; cat $OUT/ubench_mfma_pass0.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end > $OUT/bench_traced.json 2> $OUT/bench_traced.err; echo "trace rc=$?"
for k in welch fscan xcorr3; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/solo_$k -- python3 $ROOT/tools/run_kernel.py $k --reps 60 > $OUT/solo_$k.log 2>&1; echo "solo $k rc=$?"; tail -1 $OUT/solo_$k.log
done
run() { # name, counters
  timeout -k 10 200 rocprofv3 --pmc $2 --output-format csv -d $OUT/pmc/$1 -- python3 $ROOT/tools/run_kernel.py welch --reps 2 > $OUT/pmc_$1.log 2>&1; echo "pmc $1 rc=$?"
}
mkdir -p $OUT/pmc
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
run b "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES"
run c "FETCH_SIZE"
run d "WRITE_SIZE"
timeout -k 10 100 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc/calib -- $ROOT/tools/calib_fetch > $OUT/pmc_calib.log 2>&1; echo "calib rc=$?"
find $OUT -name "*stats*.csv" | head
