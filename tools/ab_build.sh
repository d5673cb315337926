#!/bin/bash
# Build timing-only variants of libgpsjam_hip.so into build_ab/ (ablation / A-B experiments):
#   tools/ab_build.sh <name> "<extra hipcc flags>" [source, default k_welch.hip]
# and run them with GPSJAM_LIB=$PWD/build_ab/libgpsjam_<name>.so python tools/run_kernel.py welch
set -e
cd "$(dirname "$0")/../gps-jamming_amd/csrc"
mkdir -p ../../build_ab
make -s -j8
SRC=${3:-k_welch.hip}
OBJS=""
for o in api.o host_io.o k_scan.o k_welch.o k_xcorr.o k_synth.o k_acq.o comm.o; do
  if [ "$o" = "${SRC%.hip}.o" ]; then OBJS="$OBJS /tmp/ab_$1.o"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -I. $2 -c $SRC -o /tmp/ab_$1.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../build_ab/libgpsjam_$1.so $OBJS -ldl
echo built build_ab/libgpsjam_$1.so
