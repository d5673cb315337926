#!/usr/bin/env python3
"""How much does a kernel running beside K2 stretch it?  Two gpsjam contexts on two torch
streams over the same capture: K2 x reps on one, `--side` kernel back-to-back on the other.
    python tools/corun.py --side fscan|k1|k3|k4|copy|fill|none [--reps 20]
copy / fill: a device-to-device copy / a fill of 1 GiB (torch, next to no VALU work) -- tells HBM traffic apart
from VALU issue as the cause of the stretch."""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gps-jamming_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--side", default="fscan")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--k2-priority", type=int, default=0, help="torch stream priority of the K2 stream (-1 = high)")
    ap.add_argument("--side-priority", type=int, default=0)
    ap.add_argument("--side-per-k2", type=int, default=1, help="side launches issued per K2 launch")
    ap.add_argument("--side-cu-every", type=int, default=0,
                    help="confine the side stream to every n-th CU (hipExtStreamCreateWithCUMask); 0 = no mask")
    args = ap.parse_args()
    import torch
    import gpsjam
    from gpsjam.synth import StreamSpec

    nbytes = 1 << 30
    ns = nbytes // 2
    s1, s2 = torch.cuda.Stream(priority=args.k2_priority), torch.cuda.Stream(priority=args.side_priority)
    if args.side_cu_every:
        import ctypes as C
        hip = C.CDLL("libamdhip64.so")
        words = (C.c_uint32 * 8)()
        for cu in range(0, 256, args.side_cu_every):
            words[cu // 32] |= 1 << (cu % 32)
        h = C.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(h), 8, words)
        assert rc == 0, rc
        s2 = torch.cuda.ExternalStream(h.value)
    a, b = gpsjam.Device(0), gpsjam.Device(0)
    a.set_stream(s1.cuda_stream)
    b.set_stream(s2.cuda_stream)
    cap = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    with torch.cuda.stream(s1):
        a.synth_dev(StreamSpec(seed=1234, jam_start=int(0.4 * ns), jam_end=int(0.7 * ns), jam_sigma=60.0), ns, cap)
    rows = a.welch_rows(nbytes, 2048000, 4096)
    nch = a.chunk_count(nbytes, 65536)
    psd = torch.empty((rows, 4096), dtype=torch.float32, device="cuda")
    pw = torch.empty(nch, dtype=torch.float32, device="cuda")
    amp = torch.zeros(4, dtype=torch.int64, device="cuda")
    on = torch.zeros(4, dtype=torch.int64, device="cuda")
    dst = torch.empty(nbytes, dtype=torch.uint8, device="cuda") if args.side in ("copy", "fill") else None
    torch.cuda.synchronize()

    def side():
        if args.side == "copy":
            with torch.cuda.stream(s2):
                dst.copy_(cap)
            return
        if args.side == "fill":
            with torch.cuda.stream(s2):
                dst.fill_(7)
            return
        if args.side == "k1small":      # the same bytes per K2 launch, but out of a 16-MiB window: L2 / Infinity Cache, no HBM
            for _ in range(64):
                b.chunk_power_dev(cap, 1 << 24, 65536, pw)
            return
        if args.side == "k1x8":         # 8 launches over consecutive 128-MiB windows: HBM, same launch count ballpark
            for k in range(8):
                b.chunk_power_dev(cap[k << 27:], 1 << 27, 65536, pw[k << 11:])
            return
        if args.side == "fscan":
            b.stream_scan_dev(cap, nbytes, 65536, pw, 0.0, amp, 200000, 1000, 50.0, on)
        elif args.side == "k1":
            b.chunk_power_dev(cap, nbytes, 65536, pw)
        elif args.side == "k3":
            b.amp_stats_dev(cap, nbytes, 0.0, amp)
        elif args.side == "k4":
            b.onset_dev(cap, nbytes, 200000, 1000, 50.0, on)

    for _ in range(30):   # settle clocks
        a.welch_dev(cap, nbytes, 2048000, 4096, 2.048e6, psd)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s1)
    f0.record(s2)
    for _ in range(args.reps):
        a.welch_dev(cap, nbytes, 2048000, 4096, 2.048e6, psd)
        if args.side != "none":
            for _ in range(args.side_per_k2):
                side()
    e1.record(s1)
    f1.record(s2)
    torch.cuda.synchronize()
    print(f"side={args.side:5s} K2 {e0.elapsed_time(e1) / args.reps:.3f} ms per launch; "
          f"side stream busy {f0.elapsed_time(f1) / args.reps:.3f} ms per K2 launch")


if __name__ == "__main__":
    main()
