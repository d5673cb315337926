#!/usr/bin/env python3
"""Run ONE kernel family of the hot path repeatedly on a synthetic capture resident in HBM --
the thing to put after `rocprofv3 ... --` for kernel-trace and PMC passes.

    python tools/run_kernel.py welch --reps 5 [--nperseg 4096] [--bytes 1073741824]
    python tools/run_kernel.py scan  --reps 5      (K1 + threshold + K3 + K4, one kernel family each)
    python tools/run_kernel.py fscan --reps 5      (gj_stream_scan_dev: fused pass + tail)
    python tools/run_kernel.py cscan --reps 5 [--slice 524288]   (gj_capture_scan_dev: + threshold and TDOA slot in the tail)
    python tools/run_kernel.py xcorr --reps 5
    python tools/run_kernel.py acq   --reps 5      (one cold acquisition search per repetition)
Prints the average wall time per repetition measured with HIP events on the launch stream.
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "gps-jamming_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["welch", "scan", "fscan", "cscan", "thr", "xcorr", "xcorr3", "k1", "k3", "k4", "acq"])
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--nperseg", type=int, default=4096)
    ap.add_argument("--bytes", type=int, default=1 << 30)
    ap.add_argument("--slice", type=int, default=1 << 19)
    args = ap.parse_args()
    import numpy as np
    import gpsjam
    from gpsjam.synth import StreamSpec

    dev = gpsjam.Device(0)
    nbytes, ns = args.bytes, args.bytes // 2
    cap = dev.alloc(nbytes)
    spec = StreamSpec(seed=1234, antenna=0, jam_start=int(0.4 * ns), jam_end=int(0.7 * ns), jam_sigma=60.0)
    dev.synth_dev(spec, ns, cap)
    rows = dev.welch_rows(nbytes, 2048000, args.nperseg)
    nch = dev.chunk_count(nbytes, 65536)
    d_psd = dev.alloc(4 * max(rows, 1) * args.nperseg)
    d_pow, d_stats, d_mask = dev.alloc(4 * nch), dev.alloc(12), dev.alloc(nch)
    d_amp, d_on = dev.alloc(32), dev.alloc(32)
    d_slot = dev.alloc(dev.tdoa_slot_bytes(args.slice))
    d_starts, d_lags, d_peaks = dev.alloc(16), dev.alloc(4), dev.alloc(4)
    d_starts.upload(np.array([int(0.4 * ns), int(0.4 * ns) + 3], np.int64))
    d_starts3, d_lags3, d_peaks3 = dev.alloc(24), dev.alloc(12), dev.alloc(12)
    d_starts3.upload(np.array([int(0.4 * ns), int(0.4 * ns) + 3, int(0.4 * ns) - 5], np.int64))

    srch = None
    if args.what == "acq":            # 32 PRNs x 71 Doppler bins x 10 ms on the quiet start of the capture
        from gpsjam.gnss import AcqSearch
        srch = AcqSearch(dev)

    def once():
        if args.what == "acq":
            srch.search_dev(cap, nbytes, 0)
        elif args.what == "welch":
            dev.welch_dev(cap, nbytes, 2048000, args.nperseg, 2.048e6, d_psd)
        elif args.what == "k1":
            dev.chunk_power_dev(cap, nbytes, 65536, d_pow)
        elif args.what == "k3":
            dev.amp_stats_dev(cap, nbytes, 0.0, d_amp)
        elif args.what == "k4":
            dev.onset_dev(cap, nbytes, 200000, 1000, 50.0, d_on)
        elif args.what == "fscan":
            dev.stream_scan_dev(cap, nbytes, 65536, d_pow, 0.0, d_amp, 200000, 1000, 50.0, d_on)
        elif args.what == "cscan":
            dev.capture_scan_dev(cap, nbytes, 65536, d_pow, 0.0, d_amp, 200000, 1000, 50.0, d_on, d_stats=d_stats, d_mask=d_mask,
                                 slice_samples=args.slice, d_slot=d_slot)
        elif args.what == "thr":
            dev.power_threshold_dev(d_pow, nch, d_stats, d_mask)
        elif args.what == "scan":
            dev.chunk_power_dev(cap, nbytes, 65536, d_pow)
            dev.power_threshold_dev(d_pow, nch, d_stats, d_mask)
            dev.amp_stats_dev(cap, nbytes, 0.0, d_amp)
            dev.onset_dev(cap, nbytes, 200000, 1000, 50.0, d_on)
        elif args.what == "xcorr3":   # BASELINE configs[3]: 3 antennas, pairs (0,1), (0,2), (1,2), N = 2^19
            dev.xcorr_lags_dev([cap, cap, cap], [nbytes] * 3, d_starts3, 1 << 19, [(0, 1), (0, 2), (1, 2)],
                               d_lags3, d_peaks3)
        else:
            dev.xcorr_lags_dev([cap, cap], [nbytes, nbytes], d_starts, 1 << 19, [(0, 1)], d_lags, d_peaks)

    once()
    dev.synchronize()
    dev.timer_start()
    for _ in range(args.reps):
        once()
    ms = dev.timer_stop() / args.reps
    extra = ""
    if args.what == "k4":
        extra = f" onset={d_on.download(np.int64, 1)[0]}"
    if args.what == "xcorr":
        extra = f" lag={d_lags.download(np.int32)[0]}"
    if args.what == "xcorr3":
        extra = f" lags={d_lags3.download(np.int32).tolist()} (195 MiB four-step floor -> {195 * 1.048576 / ms:.0f} GB/s)"
    print(f"{args.what}: {ms:.4f} ms per repetition over {nbytes} bytes -> {nbytes / ms / 1e6:.1f} GB/s algorithmic{extra}")
    dev.close()


if __name__ == "__main__":
    main()
