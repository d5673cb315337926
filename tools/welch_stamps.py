#!/usr/bin/env python3
"""Phase breakdown of welch_kernel<4096> from the s_memtime-stamped diagnostic build:
    tools/ab_build.sh stamps -DGJ_STAMPS
    GPSJAM_LIB=$PWD/build_ab/libgpsjam_stamps.so python tools/welch_stamps.py
(shares, not durations: the stamps drain LDS reads and forbid overlaps the real kernel has)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gps-jamming_amd"))
import numpy as np   # noqa: E402
import gpsjam        # noqa: E402
from gpsjam.synth import StreamSpec   # noqa: E402

dev = gpsjam.Device(0)
lib = gpsjam._ffi.load()
nbytes = 1 << 30
cap = dev.alloc(nbytes)
dev.synth_dev(StreamSpec(seed=1), nbytes // 2, cap)
rows = dev.welch_rows(nbytes, 2048000, 4096)
psd = dev.alloc(4 * rows * 4096)
dev.welch_dev(cap, nbytes, 2048000, 4096, 2.048e6, psd)
dev.synchronize()
out = (C.c_ulonglong * 8)()
lib.gj_debug_welch_stamps(out, 1)
dev.timer_start()
dev.welch_dev(cap, nbytes, 2048000, 4096, 2.048e6, psd)
ms = dev.timer_stop()
lib.gj_debug_welch_stamps(out, 0)
v = np.array(list(out), dtype=np.float64)
names = ["butterflies", "scatter", "barrier wait", "gather", "unpack+window (+load wait)", "detrend + |X|^2",
         "whole loop"]
steps = v[7]
print(f"kernel {ms:.3f} ms; {steps:.0f} wave-steps")
for n, x in zip(names, v[:7]):
    print(f"  {n:30s} {x / steps:9.0f} cycles per wave-step  ({100 * x / v[6]:5.1f} % of loop)")
