#!/usr/bin/env python3
"""Three 10-s capture files -> resident captures with their scan + PSD results: one after the other (what the drop-ins and
bench.py's file_to_results do) against three host threads at once.  GPSJAM_FILL_THREADS is read once per process, so
run one process per setting:   GPSJAM_FILL_THREADS=4 python tools/ingest3_probe.py"""
import os
import sys
import tempfile
import threading
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "gps-jamming_amd"))


def main():
    import numpy as np
    import gpsjam
    from gpsjam.synth import StreamSpec, generate
    n = 20_480_000
    d = tempfile.mkdtemp(dir="/tmp")
    paths = []
    for a in range(3):
        raw = generate(StreamSpec(seed=1234 + a, antenna=a, delay=(0, 3, -5)[a], jam_start=int(0.4 * n), jam_end=int(0.7 * n), jam_sigma=60.0), n)
        p = os.path.join(d, f"ant{a}.bin")
        raw.tofile(p)
        paths.append(p)
    dev = gpsjam.Device(0)

    def one(p, out, k):
        out[k] = dev.ingest(p, rssi_threshold=0.0, welch=(2048000, 1024), want_db=False)

    def sequential():
        t0 = time.perf_counter()
        held = [None] * 3
        for k, p in enumerate(paths):
            one(p, held, k)
        ms = (time.perf_counter() - t0) * 1e3
        for c in held:
            c.free()
        return ms

    def threaded():
        t0 = time.perf_counter()
        held = [None] * 3
        ts = [threading.Thread(target=one, args=(p, held, k)) for k, p in enumerate(paths)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        ms = (time.perf_counter() - t0) * 1e3
        for c in held:
            c.free()
        return ms

    def in_library():
        t0 = time.perf_counter()
        held = dev.ingest_many(paths, rssi_threshold=0.0, welch=(2048000, 1024), want_db=False)
        ms = (time.perf_counter() - t0) * 1e3
        for c in held:
            c.free()
        return ms

    for f in (sequential, threaded, in_library):
        f()
    seq = sorted(sequential() for _ in range(9))
    thr = sorted(threaded() for _ in range(9))
    lib = sorted(in_library() for _ in range(9))
    print(f"fill threads {os.environ.get('GPSJAM_FILL_THREADS', 'by size')}: one after the other min {seq[0]:.2f} med {seq[4]:.2f} max {seq[-1]:.2f} ms; "
          f"three Python threads min {thr[0]:.2f} med {thr[4]:.2f} max {thr[-1]:.2f}; gj_ingest_files min {lib[0]:.2f} med {lib[4]:.2f} max {lib[-1]:.2f}")
    for p in paths:
        os.remove(p)
    os.rmdir(d)
    dev.close()


if __name__ == "__main__":
    main()
