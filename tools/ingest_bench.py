#!/usr/bin/env python3
"""Host -> HBM ingest of a 1 GiB capture (gj_upload from a numpy array, gj_upload_file from a file in /dev/shm)
against the number of fill threads (GPSJAM_FILL_THREADS) and the file path (GPSJAM_FILE_READ=mmap|pread); the
five times in brackets are back-to-back calls, the first one being the first read of the freshly written file.
    python tools/ingest_bench.py            # the matrix, one child process per setting (the knobs are read once)
    python tools/ingest_bench.py --one      # this process's environment only
"""
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "gps-jamming_amd"))


def one():
    import numpy as np
    import gpsjam
    n = 1 << 30
    dev = gpsjam.Device(0)
    raw = np.random.RandomState(0).randint(96, 160, n, dtype=np.uint8)
    path = "/dev/shm/gpsjam_ingest_bench.bin"
    raw.tofile(path)
    try:
        out = []
        t0 = time.perf_counter()
        b = dev.alloc(n)
        b.free()
        out.append(f"malloc+free {(time.perf_counter() - t0) * 1e3:5.1f} ms")
        for name, src in (("array", raw), ("file", path)):
            best, every = 1e9, []
            for _ in range(5):
                t0 = time.perf_counter()
                cap = dev.capture(src)
                dt = time.perf_counter() - t0
                cap.free()
                best = min(best, dt)
                every.append(f"{dt * 1e3:.0f}")
            out.append(f"{name} {best * 1e3:6.1f} ms = {n / best / 1e9:5.1f} GB/s ({' '.join(every)})")
        print(f"threads={os.environ.get('GPSJAM_FILL_THREADS', '8'):>2s} {os.environ.get('GPSJAM_FILE_READ', 'mmap'):5s} "
              + "   ".join(out), flush=True)
    finally:
        os.unlink(path)
        dev.close()


def main():
    if "--one" in sys.argv:
        return one()
    for _ in range(2):
        for threads, how in ((8, "mmap"), (8, "pread"), (12, "mmap"), (12, "pread")):
            env = dict(os.environ, GPSJAM_FILL_THREADS=str(threads), GPSJAM_FILE_READ=how)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, check=True)


if __name__ == "__main__":
    main()
