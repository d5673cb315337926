#!/usr/bin/env python3
"""Timeline of the last full step in a rocprofv3 --kernel-trace CSV (which kernels overlap K2).
    python tools/trace_timeline.py gpurun_out/<dir>/<host>/<pid>_kernel_trace.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
w = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:56], r["Queue_Id"]) for r in rows)
t0 = w[0][0]
k2 = [(a - t0, b - a) for a, b, n, q in w if "welch_kernel<" in n]
print("K2 durations (us):", [round(d / 1e3) for _, d in k2])
# the last K2 launch of a pipeline step = the last one a fused scan starts beside (the solo timings that bench.py takes
# afterwards have nothing next to them)
scans = [a - t0 for a, b, n, q in w if "stream_scan_kernel" in n]
steps = [i for i, (a, d) in enumerate(k2) if any(a - 200000 <= s < a + d for s in scans)]
if len(steps) >= 3:
    i = steps[-2]
    start, stop = k2[i][0] - 50000, k2[i + 1][0] - 50000
    for a, b, n, q in w:
        if start <= a - t0 < stop:
            print(f"{(a - t0 - start) / 1e3:9.1f} us +{(b - a) / 1e3:8.1f}  q{q} {n}")
