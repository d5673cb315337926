#!/bin/bash
# kernel trace of tools/acq_bench.py: bash tools/acq_profile.sh <tag>  -> gpurun_out/<tag>/
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/acq_bench.py > $OUT/acq_bench.log 2>&1; echo "trace rc=$?"
cat $OUT/acq_bench.log | tail -3
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "acq" in r["Name"] or "fill" in r["Name"]:
        print(f'{r["Name"][:64]:64s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"]) / 1e3:8.1f} us  min {float(r["MinNs"]) / 1e3:8.1f}')
PY
