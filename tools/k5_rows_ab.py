#!/usr/bin/env python3
"""K5 variants (VERDICT r05 "next" 5), interleaved on one box: three antennas, three pairs, slices of --n samples; one child
process per (variant, round): `default` is the shipped library, `lib:<name>` an alternative build made by
tools/ab_build.sh (e.g. `tools/ab_build.sh xcnt "-DGJ_XC_NT=1" k_xcorr.hip`).  Each child prints the solve time (events
around `reps` back-to-back gj_xcorr_lags_dev calls) and a digest of lags / peaks / margins: every variant must give the
same bytes.  (Round 6 also measured a three-antenna row kernel that transforms every row once -- `pair` / `trio` in
profiles/r06_k5_rows_ab_v1.txt; it was no faster and is gone.)
    python tools/k5_rows_ab.py [--n 524288] [--rounds 3] [--variants default,lib:xcnt]"""
import argparse
import hashlib
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "gps-jamming_amd"), REPO]


def child(args):
    import numpy as np
    import gpsjam
    from gpsjam.synth import StreamSpec
    n = args.n
    with gpsjam.Device(0) as dev:
        caps = []
        for a, d in enumerate((0, 3, -5)):
            c = dev.alloc(2 * n + 64)
            dev.synth_dev(StreamSpec(seed=9, antenna=a, delay=d, jam_start=0, jam_end=1 << 40, jam_sigma=50.0), n + 32, c)
            caps.append(c)
        starts = dev.alloc(64)
        starts.upload(np.array([8, 8, 8, 0, 0, 0, 0, 0], np.int64).view(np.uint8))
        d_l, d_p, d_m = dev.alloc(64), dev.alloc(64), dev.alloc(64)
        pairs = [(0, 1), (0, 2), (1, 2)]
        call = lambda: dev.xcorr_lags_dev(caps, [2 * n + 64] * 3, starts, n, pairs, d_l, d_p, d_m)
        for _ in range(20):
            call()
        dev.synchronize()
        best, tot = 1e9, 0.0
        for _ in range(5):
            dev.timer_start()
            for _ in range(args.reps):
                call()
            ms = dev.timer_stop() / args.reps
            best, tot = min(best, ms), tot + ms
        blob = d_l.download(np.uint8, 12).tobytes() + d_p.download(np.uint8, 12).tobytes() + d_m.download(np.uint8, 12).tobytes()
        print(json.dumps({"us_avg": 1e3 * tot / 5, "us_best": 1e3 * best, "lags": d_l.download(np.int32, 3).tolist(),
                          "digest": hashlib.sha256(blob).hexdigest()[:16]}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1 << 19)
    ap.add_argument("--reps", type=int, default=100)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--variants", default="default", help="comma list: default or lib:<name> (GPSJAM_LIB=build_ab/libgpsjam_<name>.so)")
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    if args.child:
        return child(args)
    digests = set()
    for rnd in range(args.rounds):
        for variant in args.variants.split(","):
            env = dict(os.environ)
            if variant.startswith("lib:"):
                env["GPSJAM_LIB"] = os.path.join(REPO, "build_ab", f"libgpsjam_{variant[4:]}.so")
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--n", str(args.n), "--reps", str(args.reps)],
                               env=env, capture_output=True, text=True, timeout=300)
            line = next((ln for ln in r.stdout.splitlines() if ln.startswith("{")), None)
            if r.returncode or not line:
                print(f"{variant}: FAILED rc {r.returncode} {r.stderr[-300:]}", flush=True)
                continue
            d = json.loads(line)
            digests.add(d["digest"])
            print(f"round {rnd} {variant:>10}: {d['us_avg']:7.1f} us avg {d['us_best']:7.1f} us best  lags {d['lags']} digest {d['digest']}  (n = {args.n})", flush=True)
    print("results byte-equal across variants and rounds:", len(digests) == 1, flush=True)


if __name__ == "__main__":
    main()
