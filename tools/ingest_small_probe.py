#!/usr/bin/env python3
"""Ingest latency at the sizes the reference is used at (10-s = 41 MB and 60-s = 246 MB captures in /dev/shm):
host array, file read for the first time after it was written, the same file read again.  One line per case; run it
with GPSJAM_LIB pointing at different builds (and GPSJAM_FILE_READ=pread) to compare them on one box."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gps-jamming_amd"))
import numpy as np
import gpsjam

tag = sys.argv[1] if len(sys.argv) > 1 else ""
dev = gpsjam.Device(0)
for mb in (40.96, 245.76):
    n = int(mb * 1e6)
    host = np.random.RandomState(1).randint(100, 156, n, dtype=np.uint8)
    dev.ingest(host).free()
    first, again, arr = [], [], []
    for rep in range(5):
        p = os.path.join(os.environ.get("GPSJAM_PROBE_DIR", "/dev/shm"), f"gpsjam_probe_{os.getpid()}_{rep}.bin")
        host.tofile(p)
        for out in (first, again):
            t0 = time.perf_counter()
            c = dev.ingest(p)
            out.append((time.perf_counter() - t0) * 1e3)
            c.free()
        t0 = time.perf_counter()
        c = dev.ingest(host)
        arr.append((time.perf_counter() - t0) * 1e3)
        c.free()
        os.remove(p)
    f = lambda v: " ".join(f"{x:6.2f}" for x in v[1:]) + f"   median {sorted(v[1:])[len(v[1:]) // 2]:6.2f}"
    print(f"{tag:10s} {mb:7.2f} MB  host array        {f(arr)}")
    print(f"{tag:10s} {mb:7.2f} MB  file, first read  {f(first)}")
    print(f"{tag:10s} {mb:7.2f} MB  file, read again  {f(again)}", flush=True)
dev.close()
