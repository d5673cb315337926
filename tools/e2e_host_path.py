#!/usr/bin/env python3
"""End-to-end (PCIe-inclusive) rate of the host-buffer entry points on a 1 GiB numpy capture."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gps-jamming_amd"))
import numpy as np, gpsjam
dev = gpsjam.Device(0)
n = 1 << 30
raw = np.random.RandomState(0).randint(96, 160, n, dtype=np.uint8)
for name, fn in (("chunk_power", lambda: dev.chunk_power(raw)),
                 ("welch 4096", lambda: dev.welch(raw, nperseg=4096, want_db=False)),
                 ("amp_stats", lambda: dev.amp_stats(raw, 0.0))):
    fn()
    t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
    print(f"{name:12s} {dt*1e3:8.1f} ms wall for 1 GiB host buffer -> {n/2/dt/1e6:9.0f} Msamples/s, {n/dt/1e9:6.1f} GB/s; kernel {dev.last_kernel_ms:.3f} ms")
