#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Run in the build container only (``/root/reference`` must exist):

    python tests/golden/make_golden.py

Inputs are regenerated from ``gpsjam.synth`` (integer-only, bit-reproducible; their
sha256 is stored next to the outputs), written to a temp dir as ``.bin`` files and fed
to the reference's own functions:

  G1  GpsJammerApp/app/worker.py      GPSAnalysisThread.precalculate_power_profile
      GpsJammerApp/app/checkIfJamming.py  analyze_chunk_power / analyze_file_for_jamming
  G2  skrypty/widmo_plot.py:38-52     (module runs at import with a hard-coded path, so
      the nine arithmetic lines are executed here around the same
      scipy.signal.welch call the reference makes)
  G3  skrypty/triangulateRSSI.py      calculate_distance_from_file, triangulate_jammer_location
  G4  skrypty/triangulateTDOA.py      find_interference_start + scipy.signal.correlate
                                      exactly as in its __main__ (:80-89)

Only outputs (small arrays / JSON) are committed; no reference source is copied.
"""
import hashlib
import io
import json
import os
import sys
import tempfile
import types
from contextlib import redirect_stdout

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, os.path.join(REPO, "gps-jamming_amd"))
sys.path.insert(0, HERE)

from golden_inputs import (g1_stream, g2_stream, g3_streams, g4_streams,  # noqa: E402
                           G3_POSITIONS, G4_SLICES)


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def install_qt_stub():
    """worker.py:2 needs PySide6.QtCore.{QThread,Signal}; PySide6 is not installed."""
    class _Sig:
        def __init__(self, *a):
            self.emitted = []
            self.slots = []

        def connect(self, f):
            self.slots.append(f)

        def emit(self, *a):
            self.emitted.append(a)

    class Signal:
        def __init__(self, *types_):
            self.name = None

        def __set_name__(self, owner, name):
            self.name = "_sig_" + name

        def __get__(self, obj, owner=None):
            if obj is None:
                return self
            if self.name not in obj.__dict__:
                obj.__dict__[self.name] = _Sig()
            return obj.__dict__[self.name]

    class QThread:
        def __init__(self, *a, **k):
            pass

    qtcore = types.ModuleType("PySide6.QtCore")
    qtcore.QThread = QThread
    qtcore.Signal = Signal
    pkg = types.ModuleType("PySide6")
    pkg.QtCore = qtcore
    sys.modules["PySide6"] = pkg
    sys.modules["PySide6.QtCore"] = qtcore


def main():
    assert os.path.isdir(REF), "reference not present: run in the build container"
    sys.path.insert(0, os.path.join(REF, "skrypty"))
    sys.path.insert(0, os.path.join(REF, "GpsJammerApp", "app"))
    install_qt_stub()
    import importlib.util
    import triangulateRSSI as ref_rssi
    import triangulateTDOA as ref_tdoa
    import checkIfJamming as ref_cij
    spec = importlib.util.spec_from_file_location(
        "ref_worker", os.path.join(REF, "GpsJammerApp", "app", "worker.py"))
    ref_worker = importlib.util.module_from_spec(spec)
    with redirect_stdout(io.StringIO()):
        spec.loader.exec_module(ref_worker)
    from scipy import signal

    tmp = tempfile.mkdtemp(prefix="gj_golden_")
    meta = {"numpy": np.__version__, "scipy": __import__("scipy").__version__}

    # ------------------------------------------------------------------ G1
    raw = g1_stream()
    p1 = os.path.join(tmp, "g1.bin")
    raw.tofile(p1)
    with redirect_stdout(io.StringIO()):
        th = ref_worker.GPSAnalysisThread([p1])
        th.precalculate_power_profile()
    progress = [list(a) for a in th.progress_update.emitted]
    cij_pow = []
    for off in range(0, raw.size, ref_cij.CHUNK_SIZE_BYTES):
        cij_pow.append(ref_cij.analyze_chunk_power(raw[off:off + ref_cij.CHUNK_SIZE_BYTES], 0.0)[1])
    cij_pow = np.array(cij_pow, dtype=np.float64)
    cij_thr = float(np.median(cij_pow) * 4.8)
    with redirect_stdout(io.StringIO()):
        events = ref_cij.analyze_file_for_jamming(p1, cij_thr)
    np.savez(os.path.join(HERE, "g1_power.npz"),
             power_map=np.asarray(th.power_map),
             baseline=np.asarray(th.global_baseline_power),
             ranges=np.asarray(th.jamming_byte_ranges, dtype=np.int64).reshape(-1, 2),
             total_file_bytes=np.int64(th.total_file_bytes),
             total_samples=np.int64(th.total_samples),
             cij_power=cij_pow, cij_threshold=np.float64(cij_thr),
             cij_events=np.asarray(events, dtype=np.int64).reshape(-1, 2))
    meta["g1"] = {"sha256": sha(raw), "nbytes": int(raw.size), "progress": progress,
                  "power_map_dtype": str(np.asarray(th.power_map).dtype),
                  "baseline_type": type(th.global_baseline_power).__name__}

    # ------------------------------------------------------------------ G2
    raw = g2_stream()
    chunk = int(2.048e6)
    out = {}
    for nperseg in (1024, 4096):
        rows_lin, rows_db = [], []
        for off in range(0, raw.size, 2 * chunk):
            raw_chunk = raw[off:off + 2 * chunk]
            if len(raw_chunk) < nperseg * 2:
                break
            f = raw_chunk.astype(np.float32)
            i = (f[0::2] - 127.5) / 127.5
            q = (f[1::2] - 127.5) / 127.5
            z = i + 1j * q
            z = z - np.mean(z)
            _, pxx = signal.welch(z, 2.048e6, nperseg=nperseg, return_onesided=False)
            pxx = np.fft.fftshift(pxx)
            rows_lin.append(pxx)
            rows_db.append(10 * np.log10(pxx + 1e-15))
        out[f"lin_{nperseg}"] = np.array(rows_lin)
        out[f"db_{nperseg}"] = np.array(rows_db)
    np.savez(os.path.join(HERE, "g2_welch.npz"), **out)
    meta["g2"] = {"sha256": sha(raw), "nbytes": int(raw.size),
                  "dtype": str(out["lin_1024"].dtype)}

    # ------------------------------------------------------------------ G3
    raws = g3_streams()
    paths = []
    for k, r in enumerate(raws):
        p = os.path.join(tmp, f"g3_{k}.bin")
        r.tofile(p)
        paths.append(p)
    g3 = {"sha256": [sha(r) for r in raws], "distances": {}, "triangulate": {}}
    for thr in (0.0, 0.1, 0.45, 5.0):
        d = [ref_rssi.calculate_distance_from_file(p, threshold=thr, verbose=False) for p in paths]
        g3["distances"][repr(thr)] = [None if x is None else float(x) for x in d]
    for thr in (0.0, 0.1):
        with redirect_stdout(io.StringIO()):
            res = ref_rssi.triangulate_jammer_location(
                paths, antenna_positions_meters=[np.array(p) for p in G3_POSITIONS],
                reference_lat=50.06, reference_lon=19.94, tx_power=40.0, path_loss_exp=3.0,
                frequency_mhz=1575.42, threshold=thr, verbose=False)
        g3["triangulate"][repr(thr)] = json.loads(json.dumps(res, default=float))
    with redirect_stdout(io.StringIO()):
        g3["two_files_default_positions"] = json.loads(json.dumps(
            ref_rssi.triangulate_jammer_location(paths[:2], threshold=0.0), default=float))
        g3["one_file"] = ref_rssi.triangulate_jammer_location(paths[:1])
        g3["missing_file"] = json.loads(json.dumps(ref_rssi.triangulate_jammer_location(
            [paths[0], os.path.join(tmp, "nope.bin"), paths[2]], threshold=0.0), default=float))
    # amplitude statistics straight from the reference helpers
    amp_stats = {}
    for k, p in enumerate(paths):
        amp = np.abs(ref_rssi.read_iq_data(p))
        for thr in (0.0, 0.1, 0.45):
            idx = ref_rssi.find_change_point(amp, thr)
            amp_stats[f"{k}_{thr!r}"] = [None if idx is None else int(idx),
                                         None if idx is None else float(np.mean(amp[idx:]))]
    g3["amp_stats"] = amp_stats
    meta["g3"] = g3

    # ------------------------------------------------------------------ G4
    raws = g4_streams()
    g4 = {"sha256": [sha(r) for r in raws], "onset": [], "lags_own_start": {},
          "lags_common_start": {}}
    sigs = []
    for k, r in enumerate(raws):
        p = os.path.join(tmp, f"g4_{k}.bin")
        r.tofile(p)
        sigs.append(ref_tdoa.load_iq_data(p))
        g4["onset"].append(int(ref_tdoa.find_interference_start(
            sigs[-1], ref_tdoa.NOISE_SAMPLE_SIZE, ref_tdoa.DETECTION_WINDOW_SIZE,
            ref_tdoa.DETECTION_THRESHOLD_FACTOR)))
    g4["onset_short"] = int(ref_tdoa.find_interference_start(sigs[0][:200500], 200000, 1000, 50.0))
    g4["onset_none"] = int(ref_tdoa.find_interference_start(sigs[0][:250000], 200000, 1000, 50.0))
    g4["onset_alt"] = int(ref_tdoa.find_interference_start(sigs[1], 50000, 256, 20.0))
    peaks = {}
    for n in G4_SLICES:
        for (a, b) in ((0, 1), (0, 2), (1, 2)):
            for mode in ("own", "common"):
                sa = g4["onset"][a]
                sb = g4["onset"][b] if mode == "own" else g4["onset"][a]
                s_a = sigs[a][sa:sa + n]
                s_b = sigs[b][sb:sb + n]
                assert len(s_a) == n and len(s_b) == n
                c = signal.correlate(s_b, s_a, mode='full')          # (sig1, sig0)
                ac = np.abs(c)
                lag = int(np.argmax(ac) - (len(s_a) - 1))
                key = f"{n}_{a}{b}"
                g4["lags_own_start" if mode == "own" else "lags_common_start"][key] = lag
                peaks[f"{mode}_{key}"] = float(ac.max())
    g4["peaks"] = peaks
    meta["g4"] = g4

    # ------------------------------------------------------------------ G5
    # detector state machine replay: gnssdec telemetry recorded by the reference's authors
    # (GpsJammerApp/backend/helpers/wyniki/static/capture1.txt, reduced to the fields the
    # worker reads) followed by a synthetic stretch that exercises the C/N0-drop and
    # altitude branches; fed to the REFERENCE GPSAnalysisThread.process_incoming_data.
    from golden_inputs import g5_records, g5_scenario
    cap = os.path.join(REF, "GpsJammerApp", "backend", "helpers", "wyniki", "static", "capture1.txt")
    txt = open(cap, encoding="utf-8").read()
    logged = [json.loads(b[b.index("{"):]) for b in txt.split("=" * 80) if "{" in b]
    keep_pos = ("nsat", "lat", "lon", "hgt", "gdop", "clk_bias", "buffcnt")
    reduced = [{"elapsed_time": r.get("elapsed_time", 0.0),
                "position": {k: r["position"][k] for k in keep_pos if k in r.get("position", {})},
                "observations": [{k: o[k] for k in ("snr", "residual") if k in o}
                                 for o in r.get("observations", [])]} for r in logged]
    with open(os.path.join(HERE, "g5_capture1_reduced.json"), "w") as f:
        json.dump(reduced, f, separators=(",", ":"))
    records = g5_records(reduced)
    with redirect_stdout(io.StringIO()):
        th = ref_worker.GPSAnalysisThread([])
    sc = g5_scenario()
    th.power_map = sc["power_map"]
    th.global_baseline_power = sc["baseline"]
    th.jamming_byte_ranges = sc["ranges"]
    th.power_map_ready = True
    th.total_file_bytes = sc["total_file_bytes"]
    th.total_samples = th.estimated_total_samples = sc["total_file_bytes"] // 2
    with redirect_stdout(io.StringIO()):
        for r in records:
            th.process_incoming_data(r)
    g5 = {
        "n_records": len(records),
        "new_analysis_text": [a[0] for a in th.new_analysis_text.emitted],
        "new_position_data": [list(map(float, a)) for a in th.new_position_data.emitted],
        "jamming_detected_realtime": [[bool(a[0]), json.loads(json.dumps(a[1], default=float))]
                                      for a in th.jamming_detected_realtime.emitted],
        "progress_update": [[int(a[0]), a[1]] for a in th.progress_update.emitted],
        "jamming_events": json.loads(json.dumps(th.jamming_events, default=float)),
        "last_position_before_jamming": json.loads(json.dumps(th.last_position_before_jamming, default=float)),
        "final": {"jamming_detected": bool(th.jamming_detected), "median_cn0": float(th.median_cn0),
                  "current_iq_power": float(th.current_iq_power), "cn0_history_len": len(th.cn0_history)},
    }
    with open(os.path.join(HERE, "g5_replay_expected.json"), "w") as f:
        json.dump(g5, f, separators=(",", ":"), ensure_ascii=False)
    meta["g5"] = {"n_logged": len(logged), "n_records": len(records),
                  "n_events": len(th.jamming_events)}

    with open(os.path.join(HERE, "golden_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True, ensure_ascii=False)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
