#!/usr/bin/env python3
"""Time of one cold acquisition search (32 PRNs x 71 Doppler bins x 10 ms) on a quiet capture: nothing acquires, so
every PRN runs all ten integration steps (the worst case).  HIP events around 20 searches."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gps-jamming_amd"))


def main():
    import numpy as np
    import gpsjam
    from gpsjam.gnss import AcqSearch
    dev = gpsjam.Device(0)
    rng = np.random.RandomState(3)
    raw = np.clip(np.rint(rng.normal(0.0, 6.25, 2 * 2048 * 64)), -128, 127).astype(np.int16) + 128
    cap = dev.capture(raw.astype(np.uint8))
    for fs in (2.048e6, 1.024e6):
        srch = AcqSearch(dev, fs=fs)
        for _ in range(5):
            srch.search_dev(cap, cap.nbytes, 0)
        dev.synchronize()
        reps = 20
        dev.timer_start()
        for _ in range(reps):
            srch.search_dev(cap, cap.nbytes, 0)
        ms = dev.timer_stop() / reps
        n_fft = len(srch.prns) * len(srch.freqs) * srch.intg + len(srch.freqs) * srch.intg + len(srch.prns)
        found = sum(r.acquired for r in srch.results())
        print(f"fs {fs / 1e6:.3f} MS/s (FFT {2 * srch.nsamp}): {ms:.3f} ms per search, {n_fft / ms / 1e3:.1f} M transforms/s, "
              f"{found} false acquisitions", flush=True)
        srch.close()


if __name__ == "__main__":
    main()
