#!/usr/bin/env python3
"""One steady-state step out of a rocprofv3 --kernel-trace CSV, as a timeline: start relative to the step's first pack
kernel of the PREVIOUS step's end, duration, hardware queue, kernel.  A step = from one group of pack_result kernels to the next.
    python tools/step_timeline.py <..._kernel_trace.csv> [which step from the middle, default 0]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
w = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows)
packs = [i for i, x in enumerate(w) if "pack_result" in x[2] or "pack_part" in x[2]]
groups = []                       # first index of each group of pack kernels
for i in packs:
    if not groups or w[i][0] - w[groups[-1][-1]][0] > 60000:
        groups.append([i])
    else:
        groups[-1].append(i)
k = len(groups) // 2 + (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
lo, hi = groups[k][-1] + 1, groups[k + 1][-1] + 1
t0 = w[groups[k][-1]][1]
print(f"# step {k} of {len(groups)}: {(w[hi - 1][1] - t0) / 1e3:.1f} us from the end of the previous step's last pack kernel to the end of this one's")
for a, b, n, q in w[lo:hi]:
    short = n.replace("void ", "").replace("gj::", "").split("(")[0]
    print(f"{(a - t0) / 1e3:9.1f} us +{(b - a) / 1e3:7.1f}  q{q} {short}")
