#!/bin/bash
# PMC passes for the secondary kernels (VERDICT r02 missing 4): the fused scan, the K5 3-pair solve and the
# acquisition search.  Counters only, one pass per counter set, the program directly after `--`.
#   on the GPU box, from the repo root:  bash tools/pmc_secondary.sh <tag> [families...]
set -u
TAG=${1:-pmc_secondary}
shift
FAMS=${*:-"fscan xcorr3 acq"}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
REPS=4
for k in $FAMS; do
  for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "valu:SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_LDS_BANK_CONFLICT" "busy:SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    timeout -k 10 240 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/$k/$name -- python3 $ROOT/tools/run_kernel.py $k --reps $REPS > $OUT/${k}_$name.log 2>&1
    echo "pmc $k $name rc=$?"
  done
done
timeout -k 10 100 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/calib -- $ROOT/tools/calib_fetch > $OUT/calib.log 2>&1; echo "calib rc=$?"
echo $((REPS + 1)) > $OUT/launches_per_family.txt
find $OUT -name "*counter_collection.csv" | wc -l
