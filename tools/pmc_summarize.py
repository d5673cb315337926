#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc passes of tools/pmc_welch.sh into one JSON summary (averages per
dispatch of welch_kernel) with the HBM byte counts corrected as MI355X_MICROARCH.md prescribes:
FETCH_SIZE is in KiB and, on gfx950, tallies 128-B requests at 64 B for wide streaming reads;
the factor for K2's own 2-byte-per-lane pattern is calibrated by tools/calib_fetch (1 GiB read
exactly once with each pattern) in the same session.
    python tools/pmc_summarize.py gpurun_out/<dir> [measured-on-commit [nperseg]] > profiles/r02_pmc_welch/summary.json
The summary is stamped with the commit it was measured on and with a hash of the K2 sources
(bench.py compares that hash with the sources it runs on)."""
import collections
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hash():
    h = hashlib.sha256()
    for name in ("k_welch.hip", "fft_core.h"):
        with open(os.path.join(REPO, "gps-jamming_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def collect(root, kernel_substr):
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel_substr in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    root = sys.argv[1]
    k2 = collect(root, "welch_kernel")
    u16 = collect(root, "calib_read_u16").get("FETCH_SIZE")
    x4 = collect(root, "calib_read_x4").get("FETCH_SIZE")
    out = dict(sorted(k2.items()))
    gib_kib = float(1 << 20)
    corr = {"FETCH_SIZE_KiB": k2.get("FETCH_SIZE"), "WRITE_SIZE_KiB": k2.get("WRITE_SIZE")}
    if u16 and x4:
        corr["calibration"] = {"read_1GiB_u16_FETCH_SIZE_KiB": u16, "read_1GiB_x4_FETCH_SIZE_KiB": x4,
                               "factor_u16": gib_kib / u16, "factor_x4": gib_kib / x4}
    factor = (gib_kib / u16) if u16 else 2.0
    if k2.get("FETCH_SIZE") is not None:
        corr["read_bytes"] = k2["FETCH_SIZE"] * 1024.0 * factor
        corr["read_factor_applied"] = factor
    if k2.get("WRITE_SIZE") is not None:
        corr["write_bytes"] = k2["WRITE_SIZE"] * 1024.0
    if "read_bytes" in corr and "write_bytes" in corr:
        corr["hbm_bytes_per_launch"] = corr["read_bytes"] + corr["write_bytes"]
    out["_hbm_bytes_corrected"] = corr
    if k2.get("SQ_INSTS_VALU") is not None:
        out["_valu"] = {"sq_insts_valu_per_launch": k2["SQ_INSTS_VALU"],
                        "sq_insts_lds_per_launch": k2.get("SQ_INSTS_LDS"), "sq_waves": k2.get("SQ_WAVES")}
    # the commit the passes were MEASURED on: second argument, else the current HEAD
    if len(sys.argv) > 2:
        out["_commit"] = sys.argv[2]
    else:
        try:
            out["_commit"] = subprocess.run(["git", "-C", REPO, "rev-parse", "--short=12", "HEAD"], capture_output=True,
                                            text=True, check=True).stdout.strip()
        except Exception:
            out["_commit"] = "unknown"
    out["_source_hash"] = source_hash()
    nper = sys.argv[3] if len(sys.argv) > 3 else "4096"
    out["_nperseg"] = int(nper)
    out["_note"] = (f"rocprofv3 --pmc, separate passes (tools/pmc_welch.sh <dir> {nper}), welch_kernel<{nper}> on 2^30 bytes, "
                    "averages per dispatch")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
