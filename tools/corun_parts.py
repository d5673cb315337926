#!/usr/bin/env python3
"""Experiment: does K2 stop paying for the scan's HBM traffic when both walk the capture in parts that fit the
256-MB Infinity Cache?  Part i is scanned first (HBM -> also fills the cache), K2 of part i follows it and the scan
of part i+1 runs beside it.  Timing only: the per-part scan results are not merged here.
    python tools/corun_parts.py [--chunks-per-part 32] [--reps 20]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gps-jamming_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks-per-part", type=int, default=32)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--order", default="scan-first", choices=("scan-first", "k2-first"))
    args = ap.parse_args()
    import torch
    import gpsjam
    from gpsjam.synth import StreamSpec

    nbytes = 1 << 30
    ns = nbytes // 2
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
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
    part_bytes = args.chunks_per_part * 4096000
    parts = [(o, min(part_bytes, nbytes - o)) for o in range(0, nbytes, part_bytes)]
    a.reserve(a.welch_workspace(nbytes, 2048000, 4096))
    torch.cuda.synchronize()

    def whole():
        b.stream_scan_dev(cap, nbytes, 65536, pw, 0.0, amp, 200000, 1000, 50.0, on)
        a.welch_dev(cap, nbytes, 2048000, 4096, 2.048e6, psd)

    def in_parts():
        row = 0
        for off, ln in parts:
            r = a.welch_rows(ln, 2048000, 4096)
            if args.order == "scan-first":
                b.stream_scan_dev(cap[off:], ln, 65536, pw[off // 65536:], 0.0, amp, 200000, 1000, 50.0, on)
                ev = torch.cuda.Event()
                ev.record(s2)
                s1.wait_event(ev)
                if r:
                    a.welch_dev(cap[off:], ln, 2048000, 4096, 2.048e6, psd[row:])
            else:
                if r:
                    a.welch_dev(cap[off:], ln, 2048000, 4096, 2.048e6, psd[row:])
                ev = torch.cuda.Event()
                ev.record(s1)
                s2.wait_event(ev)
                b.stream_scan_dev(cap[off:], ln, 65536, pw[off // 65536:], 0.0, amp, 200000, 1000, 50.0, on)
            row += r

    def k2_parts_only():
        row = 0
        for off, ln in parts:
            r = a.welch_rows(ln, 2048000, 4096)
            if r:
                a.welch_dev(cap[off:], ln, 2048000, 4096, 2.048e6, psd[row:])
            row += r

    for name, fn in (("whole capture, scan beside K2", whole), ("K2 alone in parts", k2_parts_only),
                     (f"parts of {args.chunks_per_part} chunks, {args.order}", in_parts), ("whole capture, scan beside K2", whole)):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = __import__("time").perf_counter()
        e0.record(s1)
        for _ in range(args.reps):
            fn()
            # the next repetition's scan must not start before this repetition's K2 has finished
            ev = torch.cuda.Event()
            ev.record(s1)
            s2.wait_event(ev)
        e1.record(s1)
        torch.cuda.synchronize()
        wall = (__import__("time").perf_counter() - t0) / args.reps * 1e3
        print(f"{name:46s} {e0.elapsed_time(e1) / args.reps:.3f} ms per pass (wall {wall:.3f})")


if __name__ == "__main__":
    main()
