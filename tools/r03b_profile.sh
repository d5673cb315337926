#!/bin/bash
# Round-3 closing evidence on the final host code (kernels unchanged since tools/r03_profile.sh ran, so the solo
# traces and PMC summaries of that run stand):  bash tools/r03b_profile.sh <tag>
#   bench.py plain + under rocprofv3 --kernel-trace --stats, --force-exchange, --split at N = 1, rehearsals on one
#   GPU, ingest breakdown at 1 GiB and at the reference's capture sizes, drop-in latencies
set -u
TAG=${1:-r03b}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --force-exchange --no-cpu-baseline --no-end-to-end > $OUT/force_exchange.json 2> $OUT/force_exchange.err; echo "force rc=$?"
timeout -k 10 200 python3 bench.py --split --steps 10 --warmup 2 --precondition 10 > $OUT/split_n1.json 2> $OUT/split_n1.err; echo "split n1 rc=$?"
timeout -k 10 200 python3 bench.py --gpus 2 --split --backend gloo --share-gpu --steps 10 --warmup 2 --precondition 10 > $OUT/split_n2_share.json 2> $OUT/split_n2.err; echo "split n2 rc=$?"
timeout -k 10 200 python3 bench.py --gpus 2 --backend gloo --share-gpu --steps 10 --warmup 2 --precondition 10 --no-cpu-baseline > $OUT/weak_n2_share.json 2> $OUT/weak_n2.err; echo "weak n2 rc=$?"
timeout -k 10 200 python3 tools/ingest_overlap_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/ingest.txt; echo "ingest rc=$?"
timeout -k 10 100 python3 tools/ingest_small_probe.py final 2>&1 | grep -v amdgpu.ids > $OUT/ingest_small.txt; echo "ingest small rc=$?"
timeout -k 10 200 python3 tools/dropin_latency.py 1 10 60 2>&1 | grep -v "amdgpu.ids\|POWER SCAN\|GPS THREAD\|Uruchamianie" > $OUT/dropin_latency.txt; echo "latency rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end > $OUT/bench_traced.json 2> $OUT/bench_traced.err; echo "trace rc=$?"
find $OUT -name "*kernel_stats.csv" | head
