#!/usr/bin/env python3
"""One steady-state step of `bench.py --split` from a rocprofv3 --kernel-trace CSV: every kernel with its start
relative to the step's first K2, its duration and its queue; then where the second stream's chains end relative to K2.
    python tools/split_timeline.py <kernel_trace.csv> [K2 launches per step, default 3]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
per = int(sys.argv[2]) if len(sys.argv) > 2 else 3
w = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70], r["Queue_Id"]) for r in rows)
k2 = [i for i, x in enumerate(w) if "welch_kernel<" in x[2]]
# steps = groups of `per` consecutive K2 launches; take the group in the middle of the run
mid = k2[(len(k2) // 2) // per * per]
nxt = k2[(len(k2) // 2) // per * per + per]
t0 = w[mid][0]
print(f"step length {(w[nxt][0] - t0) / 1e3:.1f} us")
for a, b, n, q in w[mid:nxt]:
    print(f"{(a - t0) / 1e3:9.1f} us +{(b - a) / 1e3:8.1f}  q{q} {n}")
step = w[mid:nxt]
k2_end = max(b for a, b, n, q in step if "welch_" in n)
comb = [(a, b, n) for a, b, n, q in step if "combine_" in n or "pack_result_batch" in n]
front = [(a, b, n) for a, b, n, q in step if any(s in n for s in ("stream_scan", "onset_", "amp_", "tdoa_slot", "slots_pick", "xc_", "part_slot"))]
print(f"K2 ends at {(k2_end - t0) / 1e3:.1f} us")
if front:
    print(f"front chain (scan ... K5) ends at {(max(b for a, b, n in front) - t0) / 1e3:.1f} us")
if comb:
    print(f"combine (assemble, statistics, pack: {len(comb)} launches) runs {(min(a for a, b, n in comb) - t0) / 1e3:.1f} .. "
          f"{(max(b for a, b, n in comb) - t0) / 1e3:.1f} us -- the PREVIOUS step's part vectors, under this step's K2"
          if max(b for a, b, n in comb) <= k2_end else
          f"combine ({len(comb)} launches) runs {(min(a for a, b, n in comb) - t0) / 1e3:.1f} .. {(max(b for a, b, n in comb) - t0) / 1e3:.1f} us: "
          f"ends {(max(b for a, b, n in comb) - k2_end) / 1e3:.1f} us AFTER this step's K2")
