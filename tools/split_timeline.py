#!/usr/bin/env python3
"""One steady-state step of `bench.py --split` from a rocprofv3 --kernel-trace CSV: every kernel with its start
relative to the step's first K2, its duration and its queue.
    python tools/split_timeline.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
w = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70], r["Queue_Id"]) for r in rows)
k2 = [i for i, x in enumerate(w) if "welch_kernel<" in x[2]]
# steps = groups of 3 consecutive K2 launches; take the group in the middle of the run
mid = k2[(len(k2) // 2) // 3 * 3]
nxt = k2[(len(k2) // 2) // 3 * 3 + 3]
t0 = w[mid][0]
print(f"step length {(w[nxt][0] - t0) / 1e3:.1f} us")
for a, b, n, q in w[mid:nxt]:
    print(f"{(a - t0) / 1e3:9.1f} us +{(b - a) / 1e3:8.1f}  q{q} {n}")
