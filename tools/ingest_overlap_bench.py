#!/usr/bin/env python3
"""Where the time of an overlapped ingest goes: 1-GiB host buffer -> HBM with (a) plain gj_upload, (b) gj_ingest with
nothing to compute, (c) the fused scan only, (d) K2 only, (e) both; wall clock of the C call and of the Python call."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gps-jamming_amd"))
import numpy as np   # noqa: E402
import gpsjam        # noqa: E402
from gpsjam.synth import StreamSpec   # noqa: E402

dev = gpsjam.Device(0)
nbytes = 1 << 30
d = dev.alloc(nbytes)
dev.synth_dev(StreamSpec(seed=1234, jam_start=int(0.4 * nbytes / 2), jam_end=int(0.7 * nbytes / 2), jam_sigma=60.0), nbytes // 2, d)
host = d.download(np.uint8, nbytes)
d.free()


def timed(fn, reps=4):
    out = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = fn()
        out.append((time.perf_counter() - t0) * 1e3)
        ms = getattr(r, "ingest_ms", None)
        r.free()
    return min(out), sum(out) / len(out), ms


cases = {
    "gj_upload": lambda: dev.capture(host),
    "ingest, nothing": lambda: dev.ingest(host, chunk_bytes=0),
    "ingest, scan": lambda: dev.ingest(host),
    "ingest, K2": lambda: dev.ingest(host, chunk_bytes=0, welch=(2048000, 4096)),
    "ingest, scan + K2": lambda: dev.ingest(host, welch=(2048000, 4096)),
}
for name, fn in cases.items():
    fn().free()
for rnd in range(2):
    for name, fn in cases.items():
        lo, avg, ms = timed(fn)
        print(f"{name:20s} min {lo:7.2f} ms  avg {avg:7.2f} ms  (C call: upload {ms[0]:.2f} / total {ms[1]:.2f} ms)" if ms else
              f"{name:20s} min {lo:7.2f} ms  avg {avg:7.2f} ms", flush=True)
dev.close()
