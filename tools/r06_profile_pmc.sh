#!/bin/bash
# Round-6 evidence:  bash tools/r06_profile.sh <tag>     (on the GPU box, from the repo root)
#   bench.py plain + under rocprofv3 --kernel-trace --stats; K2 solo at 4096 and 1024 from ONE box, back to back; the
#   per-capture chain (gj_capture_scan_dev) solo at 1 GiB and at the reference's 10-s size; the deployment step (graph /
#   eager) + its kernel trace; the split path as rank 0 of eight (emulated) + trace; PMC for welch 4096 / 1024, the
#   capture scan and K5 (their sources changed this round)
set -u
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
bash tools/pmc_welch.sh $TAG/pmc_welch4096 4096 > $OUT/pmc_welch4096.log 2>&1; echo "pmc welch4096 rc=$?"
bash tools/pmc_welch.sh $TAG/pmc_welch1024 1024 > $OUT/pmc_welch1024.log 2>&1; echo "pmc welch1024 rc=$?"
bash tools/pmc_secondary.sh $TAG/pmc_sec "cscan xcorr3" > $OUT/pmc_sec.log 2>&1; echo "pmc sec rc=$?"
find $OUT -name "*kernel_stats.csv" | head -20
