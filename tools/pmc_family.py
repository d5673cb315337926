#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tools/pmc_secondary.sh for one kernel family: counters per
REPETITION (one fused scan / one 3-pair K5 solve / one acquisition search = several kernels) and per kernel,
HBM bytes corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE is in KiB and, on gfx950, counts a 128-B
request as 64 B for wide streaming reads; the factor is calibrated in the same session by tools/calib_fetch,
which reads 1 GiB exactly once with 2-byte and with 16-byte loads).
    python tools/pmc_family.py gpurun_out/<tag> fscan  [commit] > profiles/r03_pmc_scan/summary.json
    python tools/pmc_family.py gpurun_out/<tag> xcorr3 [commit] > profiles/r03_pmc_xcorr/summary.json
    python tools/pmc_family.py gpurun_out/<tag> acq    [commit] > profiles/r03_pmc_acq/summary.json
Stamped with the commit measured on and a hash of the family's sources (bench.py compares it)."""
import collections
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FAMILIES = {
    # run_kernel.py mode -> (kernel-name substrings of the family, sources hashed)
    "fscan": (("stream_scan_kernel", "scan_tail_kernel", "amp_finalize", "onset_", "chunk_power_finalize"), ("k_scan.hip",)),
    # round 5: what a pipeline step launches per capture (gj_capture_scan_dev: fused pass + tail with threshold; the slot
    # kernel only for slices beyond the tail's own limit)
    "cscan": (("stream_scan_kernel", "scan_tail_kernel", "tdoa_slot_kernel", "chunk_power_finalize"), ("k_scan.hip",)),
    "xcorr3": (("xc_",), ("k_xcorr.hip", "fft_core.h")),
    "acq": (("acq_",), ("k_acq.hip", "fft_core.h")),
    "welch": (("welch_",), ("k_welch.hip", "fft_core.h")),
}


def source_hash(names):
    h = hashlib.sha256()
    for name in names:
        with open(os.path.join(REPO, "gps-jamming_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def short(name):
    name = name.split("(")[0]
    return name.replace("void ", "").replace("gj::", "").strip()


def collect(root, substrs):
    """{counter: {kernel: [values per dispatch]}} over every pass directory under root"""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if any(s in r["Kernel_Name"] for s in substrs):
                agg[r["Counter_Name"]][short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return agg


def main():
    root, fam = sys.argv[1], sys.argv[2]
    substrs, sources = FAMILIES[fam]
    launches = int(open(os.path.join(root, "launches_per_family.txt")).read().split()[0])
    agg = collect(os.path.join(root, fam), substrs)
    cal = collect(os.path.join(root, "calib"), ("calib_read",))
    u16 = cal.get("FETCH_SIZE", {}).get("calib_read_u16")
    x4 = cal.get("FETCH_SIZE", {}).get("calib_read_x4")
    gib_kib = float(1 << 20)
    f_u16 = gib_kib / (sum(u16) / len(u16)) if u16 else None
    f_x4 = gib_kib / (sum(x4) / len(x4)) if x4 else None
    factor = f_x4 or 2.0                 # these kernels read with 8- and 16-byte loads
    per_rep = {c: sum(sum(v) for v in ks.values()) / launches for c, ks in agg.items()}
    per_kernel = {}
    for c, ks in agg.items():
        for k, v in ks.items():
            per_kernel.setdefault(k, {"dispatches_per_repetition": len(v) / launches})[c + "_per_dispatch"] = sum(v) / len(v)
    out = {"family": fam, "kernels": sorted(per_kernel), "repetitions_measured": launches,
           "per_repetition": dict(sorted(per_rep.items())), "per_kernel": per_kernel}
    corr = {"calibration": {"factor_u16": f_u16, "factor_x4": f_x4}, "read_factor_applied": factor}
    if "FETCH_SIZE" in per_rep:
        corr["read_bytes"] = per_rep["FETCH_SIZE"] * 1024.0 * factor
    if "WRITE_SIZE" in per_rep:
        corr["write_bytes"] = per_rep["WRITE_SIZE"] * 1024.0
    if "read_bytes" in corr and "write_bytes" in corr:
        corr["hbm_bytes_per_repetition"] = corr["read_bytes"] + corr["write_bytes"]
    out["_hbm_bytes_corrected"] = corr
    if len(sys.argv) > 3:
        out["_commit"] = sys.argv[3]
    else:
        try:
            out["_commit"] = subprocess.run(["git", "-C", REPO, "rev-parse", "--short=12", "HEAD"], capture_output=True,
                                            text=True, check=True).stdout.strip()
        except Exception:
            out["_commit"] = "unknown"
    out["_source_hash"] = source_hash(sources)
    out["_sources"] = list(sources)
    out["_note"] = ("rocprofv3 --pmc, one pass per counter set (tools/pmc_secondary.sh), tools/run_kernel.py " + fam +
                    " on a 2^30-byte capture; per_repetition = all kernels of one call")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
