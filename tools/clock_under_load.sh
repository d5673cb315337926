#!/bin/bash
# Shader clock while K2 / the fused scan / nothing runs (tools/clock_probe.hip beside tools/run_kernel.py, two processes)
cd ${GRAFT_REPO_ROOT:-$PWD}
tools/clock_probe idle
for w in welch fscan; do
  python3 tools/run_kernel.py $w --reps 1500 > /tmp/load_$w.txt 2>&1 &
  LP=$!
  sleep 4     # library load, capture synthesis, transient clocks
  tools/clock_probe $w
  wait $LP
  tail -1 /tmp/load_$w.txt
done
