#!/bin/bash
# Interleaved A/B of K2 builds over transform sizes on ONE box: bash tools/ab_welch_sizes.sh <tag> "<sizes>" <name>...
TAG=$1; SIZES=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
for round in 1 2 3; do
  for np in $SIZES; do
    for name in "$@"; do
      if [ "$name" = "ship" ]; then unset GPSJAM_LIB; else export GPSJAM_LIB=$ROOT/build_ab/libgpsjam_$name.so; fi
      printf "N=%s %s round %s: " $np $name $round >> $OUT/ab_sizes.txt
      python3 $ROOT/tools/run_kernel.py welch --reps 40 --nperseg $np | tail -1 >> $OUT/ab_sizes.txt
    done
  done
done
unset GPSJAM_LIB
cat $OUT/ab_sizes.txt
