#!/usr/bin/env python3
"""K2 for the reference's deployment -- three 10-s captures (40.96 MB each) in ONE gj_welch_batch_dev launch at nperseg
1024 -- against the number of workgroups per chunk (VERDICT r05 "next" 4).  One child process per split
(GPSJAM_W_BATCH_SPLITS is read once per process); events around `reps` back-to-back launches (transform + finalize).
    python tools/k2_batch_sweep.py [--splits 10,16,25,...] [--nperseg 1024] [--captures 3]
Prints one line per split; "planner" = what the library chooses on its own."""
import argparse
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
    nbytes = args.capture_bytes
    with gpsjam.Device(0) as dev:
        caps = []
        for a in range(args.captures):
            c = dev.alloc(nbytes)
            dev.synth_dev(StreamSpec(seed=5, antenna=a, jam_start=3_000_000, jam_end=1 << 40, jam_sigma=50.0), nbytes // 2, c)
            caps.append(c)
        rows = dev.welch_rows(nbytes, 2048000, args.nperseg)
        psds = [dev.alloc(4 * rows * args.nperseg) for _ in caps]
        for _ in range(20):
            dev.welch_batch_dev(caps, nbytes, 2048000, args.nperseg, 2.048e6, psds)
        dev.synchronize()
        best, tot = 1e9, 0.0
        for _ in range(args.rounds):
            dev.timer_start()
            for _ in range(args.reps):
                dev.welch_batch_dev(caps, nbytes, 2048000, args.nperseg, 2.048e6, psds)
            ms = dev.timer_stop() / args.reps
            best, tot = min(best, ms), tot + ms
        digest = [float(np.frombuffer(p.download(np.uint8, 4 * rows * args.nperseg).tobytes(), np.float32).sum()) for p in psds]
        print(json.dumps({"us_avg": 1e3 * tot / args.rounds, "us_best": 1e3 * best, "psd_sums": digest}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--splits", default="planner,8,12,16,20,25,30,38,51,64,76")
    ap.add_argument("--nperseg", type=int, default=1024)
    ap.add_argument("--captures", type=int, default=3)
    ap.add_argument("--capture-bytes", type=int, default=40_960_000)
    ap.add_argument("--reps", type=int, default=100)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    if args.child:
        return child(args)
    for sp in args.splits.split(","):
        env = dict(os.environ)
        env.pop("GPSJAM_W_BATCH_SPLITS", None)
        if sp != "planner":
            env["GPSJAM_W_BATCH_SPLITS"] = sp
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--nperseg", str(args.nperseg), "--captures",
                            str(args.captures), "--capture-bytes", str(args.capture_bytes), "--reps", str(args.reps), "--rounds",
                            str(args.rounds)], env=env, capture_output=True, text=True, timeout=300)
        line = next((ln for ln in r.stdout.splitlines() if ln.startswith("{")), None)
        if r.returncode or not line:
            print(f"splits {sp:>8}: FAILED rc {r.returncode} {r.stderr[-300:]}", flush=True)
            continue
        d = json.loads(line)
        print(f"splits {sp:>8}: {d['us_avg']:7.1f} us avg  {d['us_best']:7.1f} us best   (K2 + finalize, {args.captures} x {args.capture_bytes} B, nperseg {args.nperseg})", flush=True)


if __name__ == "__main__":
    main()
