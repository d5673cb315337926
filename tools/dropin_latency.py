#!/usr/bin/env python3
"""Latency of the drop-in entry points at the sizes the reference is used at (BASELINE configs[0] / [2]: 10-s
captures of 40 960 000 bytes): what a GUI user waits for.  Cold = first call of the process (library load, context,
lanes, pinned buffers); warm = the same call again on a fresh file (nothing cached in HBM).

    python tools/dropin_latency.py [seconds_per_file ...]      default: 1 10 60
"""
import os
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "gps-jamming_amd"))
sys.path.insert(1, os.path.join(REPO, "gps-jamming_amd", "skrypty"))
t_imp = time.perf_counter()
import numpy as np   # noqa: E402
import gpsjam        # noqa: E402
from gpsjam.synth import StreamSpec   # noqa: E402
t_imp = time.perf_counter() - t_imp


def make_file(path, nbytes, antenna):
    ns = nbytes // 2
    spec = StreamSpec(seed=1234, antenna=antenna, delay=(0, 3, -5)[antenna], jam_start=int(0.4 * ns), jam_end=int(0.7 * ns),
                      noise_sigma=6.25, jam_sigma=60.0 * (1.0, 0.9, 0.85)[antenna])
    dev = gpsjam.default_device()
    d = dev.alloc(nbytes)
    dev.synth_dev(spec, ns, d)
    d.download(np.uint8, nbytes).tofile(path)
    d.free()


def main():
    secs = [float(a) for a in sys.argv[1:]] or [1.0, 10.0, 60.0]
    where = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    t0 = time.perf_counter()
    gpsjam.default_device()
    t_dev = time.perf_counter() - t0
    print(f"import numpy + gpsjam {t_imp * 1e3:.0f} ms, first Device (library load, context, tables) {t_dev * 1e3:.0f} ms")
    from GpsJammerApp.app.worker import GPSAnalysisThread
    import triangulateRSSI
    import triangulateTDOA
    for s in secs:
        nbytes = int(s * 2048000) * 2
        sets = []
        for rep in range(3):                       # three sets of three antenna files: every timed call sees fresh paths
            paths = [os.path.join(where, f"gpsjam_lat_{os.getpid()}_{rep}_{a}.bin") for a in range(3)]
            for a, p in enumerate(paths):
                make_file(p, nbytes, a)
            sets.append(paths)
        try:
            rows = []
            for rep, paths in enumerate(sets):
                t0 = time.perf_counter()
                th = GPSAnalysisThread(paths)
                th.precalculate_power_profile()
                t1 = time.perf_counter()
                res = triangulateRSSI.triangulate_jammer_location(paths, threshold=0.0)
                t2 = time.perf_counter()
                caps = [triangulateTDOA.load_iq_data(p) for p in paths[:2]]
                on = [triangulateTDOA.find_interference_start(c, 200000, 1000, 50.0) for c in caps]
                t3 = time.perf_counter()
                assert th.power_map_ready and res["success"] and all(o > 0 for o in on), (res, on)
                rows.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
                gpsjam.release_resident()
            for name, k in (("worker power scan (file 1)", 0), ("RSSI solve (3 files)", 1), ("TDOA onsets (2 files)", 2)):
                print(f"{s:5.0f}-s files ({nbytes / 1e6:7.1f} MB each)  {name:28s} first {rows[0][k]:8.2f} ms   then {rows[1][k]:8.2f} / {rows[2][k]:8.2f} ms",
                      flush=True)
        finally:
            for paths in sets:
                for p in paths:
                    try:
                        os.remove(p)
                    except OSError:
                        pass


if __name__ == "__main__":
    main()
