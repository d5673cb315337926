#!/usr/bin/env python3
"""welch_mfma_kernel (pass 0 on the matrix pipe, GPSJAM_WELCH_MFMA=1) against welch_kernel<4096>: the PSD of the same
synthetic capture from both (the switch is read once per process, so each runs in a child), their largest relative
difference, and interleaved timings on this box.
    python tools/welch_mfma_ab.py [--bytes N] [--rounds R]"""
import argparse
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(nbytes, out):
    sys.path.insert(0, os.path.join(REPO, "gps-jamming_amd"))
    import numpy as np
    import gpsjam
    from gpsjam.synth import StreamSpec
    dev = gpsjam.Device(0)
    ns = nbytes // 2
    cap = dev.alloc(nbytes)
    dev.synth_dev(StreamSpec(seed=1234, antenna=0, jam_start=int(0.4 * ns), jam_end=int(0.7 * ns), jam_sigma=60.0), ns, cap)
    rows = dev.welch_rows(nbytes, 2048000, 4096)
    d_psd = dev.alloc(4 * rows * 4096)
    dev.welch_dev(cap, nbytes, 2048000, 4096, 2.048e6, d_psd)
    dev.synchronize()
    np.save(out, d_psd.download(np.float32, rows * 4096).reshape(rows, 4096))
    for _ in range(5):
        dev.welch_dev(cap, nbytes, 2048000, 4096, 2.048e6, d_psd)
    dev.synchronize()
    dev.timer_start()
    for _ in range(40):
        dev.welch_dev(cap, nbytes, 2048000, 4096, 2.048e6, d_psd)
    print(f"{dev.timer_stop() / 40:.4f}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bytes", type=int, default=1 << 30)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--child", default=None)
    args = ap.parse_args()
    if args.child:
        return child(args.bytes, args.child)
    import numpy as np
    times = {"0": [], "1": []}
    for rnd in range(args.rounds):
        for mode in ("0", "1"):
            env = dict(os.environ, GPSJAM_WELCH_MFMA=mode)
            r = subprocess.run([sys.executable, __file__, "--bytes", str(args.bytes), "--child", f"/tmp/welch_ab_{mode}.npy"],
                               env=env, capture_output=True, text=True, timeout=300)
            if r.returncode != 0:
                print(r.stderr[-3000:])
                return 1
            times[mode].append(float(r.stdout.strip().splitlines()[-1]))
    a, b = np.load("/tmp/welch_ab_0.npy"), np.load("/tmp/welch_ab_1.npy")
    keep = a > 1e-12
    rel = np.abs(b[keep] - a[keep]) / a[keep]
    print(f"rows {a.shape[0]}: largest relative PSD difference {rel.max():.3e} (mean {rel.mean():.2e}); worst bins "
          f"{np.argsort(np.abs(b - a).max(axis=0) / a.max(axis=0))[-4:]}")
    for mode, name in (("0", "welch_kernel<4096>"), ("1", "welch_mfma_kernel ")):
        print(f"{name}: " + "  ".join(f"{t:.4f}" for t in times[mode]) + f" ms   min {min(times[mode]):.4f}")
    print(f"ratio of the means: {sum(times['1']) / sum(times['0']):.4f}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
