#!/bin/bash
# Round-4 evidence:  bash tools/r04_profile.sh <tag>     (on the GPU box, from the repo root)
#   bench.py plain + under rocprofv3 --kernel-trace --stats; PMC for welch_kernel<1024>, the acquisition search and the
#   fused scan (their sources changed this round); the split path at N = 1, as rank 0 of eight (emulated) with its kernel
#   trace, the weak path's emulation; rehearsals of two ranks sharing the GPU
set -u
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --force-exchange --no-cpu-baseline --no-end-to-end --no-reference-point > $OUT/force_exchange.json 2> $OUT/force_exchange.err; echo "force rc=$?"
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --emulate-world 8 --force-exchange --no-cpu-baseline --no-end-to-end --no-reference-point > $OUT/weak_emulated8.json 2> $OUT/weak_emulated8.err; echo "weak emu8 rc=$?"
timeout -k 10 200 python3 bench.py --split --steps 10 --warmup 2 --precondition 10 > $OUT/split_n1.json 2> $OUT/split_n1.err; echo "split n1 rc=$?"
timeout -k 10 300 python3 bench.py --split --emulate-world 8 --force-exchange --steps 20 --warmup 5 --precondition 10 > $OUT/split_emulated8.json 2> $OUT/split_emulated8.err; echo "split emu8 rc=$?"
timeout -k 10 300 python3 bench.py --split --emulate-world 4 --force-exchange --steps 20 --warmup 5 --precondition 10 > $OUT/split_emulated4.json 2> $OUT/split_emulated4.err; echo "split emu4 rc=$?"
timeout -k 10 300 python3 bench.py --gpus 2 --split --backend gloo --share-gpu --steps 10 --warmup 2 --precondition 10 > $OUT/split_n2_share.json 2> $OUT/split_n2.err; echo "split n2 rc=$?"
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --share-gpu --steps 10 --warmup 2 --precondition 10 --no-cpu-baseline > $OUT/weak_n2_share.json 2> $OUT/weak_n2.err; echo "weak n2 rc=$?"
for r in 1 2 7; do timeout -k 10 300 python3 bench.py --split --emulate-world 8 --emulate-rank $r --force-exchange --steps 20 --warmup 5 --precondition 10 > $OUT/split_emulated8_rank$r.json 2> $OUT/split_emulated8_rank$r.err; echo "split emu8 rank $r rc=$?"; done
timeout -k 10 200 python3 tools/deployment_probe.py > $OUT/deployment_graph.txt 2>/dev/null; timeout -k 10 200 python3 tools/deployment_probe.py --eager > $OUT/deployment_eager.txt 2>/dev/null; echo "deployment rc=$?"
bash tools/pmc_welch.sh $TAG/pmc_welch4096 4096 > $OUT/pmc_welch4096.log 2>&1; echo "pmc welch4096 rc=$?"
bash tools/pmc_welch.sh $TAG/pmc_welch1024 1024 > $OUT/pmc_welch1024.log 2>&1; echo "pmc welch1024 rc=$?"
bash tools/pmc_secondary.sh $TAG/pmc_sec "acq fscan" > $OUT/pmc_sec.log 2>&1; echo "pmc sec rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-reference-point > $OUT/bench_traced.json 2> $OUT/bench_traced.err; echo "trace rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/split8_trace -- python3 $ROOT/bench.py --split --emulate-world 8 --force-exchange --steps 20 --warmup 5 --precondition 10 > $OUT/split8_traced.json 2> $OUT/split8_traced.err; echo "split8 trace rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/acq_trace -- python3 $ROOT/tools/acq_bench.py > $OUT/acq_bench.log 2>&1; echo "acq trace rc=$?"
find $OUT -name "*kernel_stats.csv" | head
