#!/bin/bash
# Interleaved A/B of K2 builds on ONE box (the pool's boxes differ by +-4 %): bash tools/ab_welch.sh <tag> <name>...
#   each <name> is a library built by tools/ab_build.sh (build_ab/libgpsjam_<name>.so); "ship" = the in-tree library
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
for round in 1 2 3 4; do
  for name in "$@"; do
    if [ "$name" = "ship" ]; then unset GPSJAM_LIB; else export GPSJAM_LIB=$ROOT/build_ab/libgpsjam_$name.so; fi
    printf "%s round %s: " $name $round >> $OUT/ab.txt
    python3 $ROOT/tools/run_kernel.py welch --reps 60 | tail -1 >> $OUT/ab.txt
  done
done
unset GPSJAM_LIB
cat $OUT/ab.txt
