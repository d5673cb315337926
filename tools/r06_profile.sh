#!/bin/bash
# Round-6 evidence:  bash tools/r06_profile.sh <tag>     (on the GPU box, from the repo root)
#   (PMC passes: tools/r06_profile_pmc.sh)
#   bench.py plain + under rocprofv3 --kernel-trace --stats; K2 solo at 4096 and 1024 from ONE box, back to back; the
#   per-capture chain (gj_capture_scan_dev) solo at 1 GiB and at the reference's 10-s size; gj_onset_dev alone (since round 6
#   the same pass + tail); the deployment step (graph /
#   eager) + its kernel trace; the split path as rank 0 of eight (emulated) + trace; PMC for welch 4096 / 1024, the
#   capture scan and K5 (their sources changed this round)
set -u
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --force-exchange --no-cpu-baseline --no-end-to-end --no-reference-point > $OUT/force_exchange.json 2> $OUT/force_exchange.err; echo "force rc=$?"
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --emulate-world 8 --force-exchange --no-cpu-baseline --no-end-to-end --no-reference-point > $OUT/weak_emulated8.json 2> $OUT/weak_emulated8.err; echo "weak emu8 rc=$?"
timeout -k 10 200 python3 bench.py --split --steps 10 --warmup 2 --precondition 10 > $OUT/split_n1.json 2> $OUT/split_n1.err; echo "split n1 rc=$?"
for r in 0 1 2 7; do timeout -k 10 300 python3 bench.py --split --emulate-world 8 --emulate-rank $r --force-exchange --steps 20 --warmup 5 --precondition 10 > $OUT/split_emulated8_rank$r.json 2> $OUT/split_emulated8_rank$r.err; echo "split emu8 rank $r rc=$?"; done
timeout -k 10 300 python3 bench.py --split --emulate-world 4 --force-exchange --steps 20 --warmup 5 --precondition 10 > $OUT/split_emulated4.json 2> $OUT/split_emulated4.err; echo "split emu4 rc=$?"
for i in 1 2; do timeout -k 10 200 python3 tools/deployment_probe.py 2>/dev/null; timeout -k 10 200 python3 tools/deployment_probe.py --eager 2>/dev/null; done > $OUT/deployment.txt; echo "deployment rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-reference-point > $OUT/bench_traced.json 2> $OUT/bench_traced.err; echo "trace rc=$?"
for n in 4096 1024 4096 1024; do timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/solo_welch${n}_$RANDOM -- python3 $ROOT/tools/run_kernel.py welch --reps 60 --nperseg $n 2>&1 | tail -1; done > $OUT/solo_welch.txt; echo "solo welch rc=$?"
for cfg in "1073741824 524288 big" "40960000 50000 small"; do set -- $cfg; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cscan_$3 -- python3 $ROOT/tools/run_kernel.py cscan --reps 40 --bytes $1 --slice $2 2>&1 | tail -1; done > $OUT/solo_cscan.txt; echo "solo cscan rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/solo_k4 -- python3 $ROOT/tools/run_kernel.py k4 --reps 40 2>&1 | tail -1 > $OUT/solo_k4.txt; echo "solo k4 (gj_onset_dev alone) rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/solo_xcorr3 -- python3 $ROOT/tools/run_kernel.py xcorr3 --reps 60 2>&1 | tail -1 > $OUT/solo_xcorr3.txt; echo "solo xcorr rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dep_trace -- python3 $ROOT/tools/deployment_probe.py > $OUT/dep_traced.txt 2>&1; echo "dep trace rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dep_trace_eager -- python3 $ROOT/tools/deployment_probe.py --eager > $OUT/dep_traced_eager.txt 2>&1; echo "dep eager trace rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/split8_trace -- python3 $ROOT/bench.py --split --emulate-world 8 --force-exchange --steps 20 --warmup 5 --precondition 10 > $OUT/split8_traced.json 2> $OUT/split8_traced.err; echo "split8 trace rc=$?"
find $OUT -name "*kernel_stats.csv" | head -20
