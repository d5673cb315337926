#!/usr/bin/env python3
"""K2 at nperseg 4096 and 1024 on one 1-GiB capture, kernel and finalize timed apart (gj_welch_timed_dev), in three
orders: each size alone ten times in a row, then interleaved -- is a size's time a property of the kernel or of what ran
before it?   python tools/k2_ab_probe.py"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "gps-jamming_amd"))


def main():
    import gpsjam
    from gpsjam.synth import StreamSpec
    dev = gpsjam.Device(0)
    nbytes = 1 << 30
    ns = nbytes // 2
    cap = dev.alloc(nbytes)
    dev.synth_dev(StreamSpec(seed=1234, antenna=0, jam_start=int(0.4 * ns), jam_end=int(0.7 * ns), jam_sigma=60.0), ns, cap)
    psd = {n: dev.alloc(4 * dev.welch_rows(nbytes, 2048000, n) * n) for n in (4096, 1024)}
    dev.reserve(max(dev.welch_workspace(nbytes, 2048000, n) for n in (4096, 1024)))

    def run(n):
        return dev.welch_timed_dev(cap, nbytes, 2048000, n, 2.048e6, psd[n])

    for n in (4096, 1024):
        run(n)
    fmt = lambda xs: " ".join(f"{k:.3f}+{f:.3f}" for k, f in xs)
    for n in (4096, 1024, 4096, 1024):
        print(f"alone {n:5d}:", fmt([run(n) for _ in range(10)]))
    inter = [(n, run(n)) for _ in range(10) for n in (4096, 1024)]
    for n in (4096, 1024):
        print(f"inter {n:5d}:", fmt([t for m, t in inter if m == n]))
    # the same through gj_welch_dev as a whole with the stopwatch around it (what bench.py's 'solo' figures are)
    for n in (4096, 1024):
        out = []
        for _ in range(8):
            dev.synchronize()
            dev.timer_start()
            dev.welch_dev(cap, nbytes, 2048000, n, 2.048e6, psd[n])
            out.append(dev.timer_stop())
        print(f"whole {n:5d}:", " ".join(f"{t:.3f}" for t in out))
    dev.close()


if __name__ == "__main__":
    main()
