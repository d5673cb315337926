#!/bin/bash
# Round-3 evidence run (on the GPU box, from the repo root):  bash tools/r03_profile.sh <tag>
#   1. bench.py plain (the JSON line)                         -> gpurun_out/<tag>/bench.json
#   2. rocprofv3 --kernel-trace --stats of the same command   -> gpurun_out/<tag>/bench_trace/
#   3. settled-clock SOLO traces of the reported kernels      -> gpurun_out/<tag>/solo_<kernel>/
#   4. PMC passes of K2 (separate runs, counters only)        -> gpurun_out/<tag>/pmc/
#   5. PMC passes of the secondary kernels                    -> gpurun_out/<tag>/pmc2/
#   6. bench.py --split at N = 1 and rehearsed on one GPU     -> gpurun_out/<tag>/split_n*.json
#   7. overlapped ingest breakdown                            -> gpurun_out/<tag>/ingest.txt
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout -k 10 200 python3 bench.py --split --steps 10 --warmup 2 --precondition 10 > $OUT/split_n1.json 2> $OUT/split_n1.err; echo "split n1 rc=$?"
timeout -k 10 200 python3 bench.py --gpus 2 --split --backend gloo --share-gpu --steps 10 --warmup 2 --precondition 10 > $OUT/split_n2_share.json 2> $OUT/split_n2.err; echo "split n2 rc=$?"
timeout -k 10 200 python3 bench.py --gpus 2 --backend gloo --share-gpu --steps 10 --warmup 2 --precondition 10 --no-cpu-baseline > $OUT/weak_n2_share.json 2> $OUT/weak_n2.err; echo "weak n2 rc=$?"
timeout -k 10 200 python3 tools/ingest_overlap_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/ingest.txt; echo "ingest rc=$?"
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
cd $ROOT && bash tools/pmc_secondary.sh $TAG/pmc2 > $OUT/pmc2.log 2>&1; echo "pmc2 rc=$?"
find $OUT -name "*stats*.csv" | head
