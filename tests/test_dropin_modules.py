"""The drop-in modules (gps-jamming_amd/skrypty, gps-jamming_amd/GpsJammerApp/app) against the
vectors captured from the reference.

Every test runs twice: with ``backend='oracle'`` on the CPU (host logic only: the device is
replaced by tests/fake_device.OracleDevice) and, marked ``gpu``, with the real library
(``backend='hip'``), which is the actual parity claim."""
import io
import json
import os
import sys
from contextlib import redirect_stdout

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(os.path.dirname(HERE), "gps-jamming_amd")
for p in (os.path.join(PKG, "skrypty"), os.path.join(PKG, "GpsJammerApp", "app")):
    if p not in sys.path:
        sys.path.insert(0, p)

import gpsjam                      # noqa: E402
import golden_inputs as gi         # noqa: E402


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def backend(request, monkeypatch):
    if request.param == "oracle":
        from fake_device import OracleDevice
        fake = OracleDevice()
        monkeypatch.setattr(gpsjam, "default_device", lambda: fake)
    else:
        monkeypatch.setattr(gpsjam, "_default", None)
    return request.param


def write_files(tmp_path, raws, prefix):
    paths = []
    for k, r in enumerate(raws):
        p = tmp_path / f"{prefix}{k}.bin"
        r.tofile(p)
        paths.append(str(p))
    return paths


def quiet(fn, *a, **k):
    with redirect_stdout(io.StringIO()):
        return fn(*a, **k)


# ------------------------------------------------------------------ triangulateRSSI
def test_rssi_module_surface():
    import triangulateRSSI as m
    for name in ("read_iq_data", "find_change_point", "meters_to_geographic_degrees",
                 "calculate_distance_from_file", "perform_grid_search", "triangulate_jammer_location",
                 "DEFAULT_CALIBRATED_TX_POWER", "DEFAULT_CALIBRATED_PATH_LOSS_EXPONENT",
                 "DEFAULT_SIGNAL_FREQUENCY_MHZ", "DEFAULT_SIGNAL_THRESHOLD", "GRID_DENSITY",
                 "SEARCH_RANGE_MULTIPLIER", "METERS_PER_DEGREE_LAT", "METERS_PER_DEGREE_LON"):
        assert hasattr(m, name), name
    import inspect
    sig = inspect.signature(m.triangulate_jammer_location)
    assert list(sig.parameters) == ["file_paths", "antenna_positions_meters", "reference_lat", "reference_lon",
                                    "tx_power", "path_loss_exp", "frequency_mhz", "threshold", "verbose"]
    assert sig.parameters["reference_lat"].default == 50.00898 and sig.parameters["threshold"].default == 0.1


def test_rssi_triangulation(backend, golden_meta, g3_raws, tmp_path):
    import triangulateRSSI as m
    g3 = golden_meta["g3"]
    paths = write_files(tmp_path, g3_raws, "ant")
    for thr_s, want in g3["distances"].items():
        got = [quiet(m.calculate_distance_from_file, p, threshold=float(thr_s), verbose=False) for p in paths]
        for a, b in zip(got, want):
            assert (a is None) == (b is None)
            if b is not None:
                np.testing.assert_allclose(a, b, rtol=1e-5)
    for thr_s, want in g3["triangulate"].items():
        got = quiet(m.triangulate_jammer_location, paths, antenna_positions_meters=[np.array(p) for p in gi.G3_POSITIONS],
                    reference_lat=50.06, reference_lon=19.94, tx_power=40.0, path_loss_exp=3.0,
                    frequency_mhz=1575.42, threshold=float(thr_s), verbose=False)
        assert set(got) == set(want)
        assert got["success"] is True and got["num_antennas"] == want["num_antennas"]
        assert got["location_meters"] == want["location_meters"]
        assert got["message"] == want["message"]
        np.testing.assert_allclose(got["distances"], want["distances"], rtol=1e-5)
        assert set(got["location_geographic"]) == set(want["location_geographic"])
        for k, v in want["location_geographic"].items():
            np.testing.assert_allclose(got["location_geographic"][k], v, rtol=1e-12)
    two = quiet(m.triangulate_jammer_location, paths[:2], threshold=0.0)
    assert two["location_meters"] == g3["two_files_default_positions"]["location_meters"]
    assert quiet(m.triangulate_jammer_location, paths[:1]) == g3["one_file"]
    miss = quiet(m.triangulate_jammer_location, [paths[0], str(tmp_path / "nope.bin"), paths[2]], threshold=0.0)
    assert miss["distances"][1] is None and miss["num_antennas"] == g3["missing_file"]["num_antennas"]
    assert miss["location_meters"] == g3["missing_file"]["location_meters"]
    none = quiet(m.triangulate_jammer_location, paths, threshold=5.0)
    assert none["success"] is False and none["distances"] == [None, None, None]
    assert none["message"].endswith("Sukcesy: 0")


# ------------------------------------------------------------------ triangulateTDOA
def test_tdoa_module(backend, golden_meta, g4_raws, tmp_path):
    import triangulateTDOA as m
    g4 = golden_meta["g4"]
    paths = write_files(tmp_path, g4_raws, "tdoa")
    caps = [m.load_iq_data(p) for p in paths]
    starts = [m.find_interference_start(c, m.NOISE_SAMPLE_SIZE, m.DETECTION_WINDOW_SIZE, m.DETECTION_THRESHOLD_FACTOR)
              for c in caps]
    assert starts == g4["onset"]
    assert m.find_interference_start(caps[0][:200500], 200000, 1000, 50.0) == -1
    n = m.CORRELATION_SLICE_SIZE
    for a, b in ((0, 1), (0, 2), (1, 2)):
        lag, _ = m.correlation_lag(caps[b][starts[b]:starts[b] + n], caps[a][starts[a]:starts[a] + n])
        assert lag == g4["lags_own_start"][f"{n}_{a}{b}"]
    # the array view of a capture is the reference's complex64 expansion
    z = np.asarray(caps[0][10:14])
    assert z.dtype == np.complex64 and z[0] == (float(g4_raws[0][20]) - 127.5) + 1j * (float(g4_raws[0][21]) - 127.5)
    out = io.StringIO()
    with redirect_stdout(out):
        rc = m.main(paths[0], paths[1])
    text = out.getvalue()
    assert f"na próbce: {starts[0]}" in text and "przy przesunięciu -4 próbek" in text
    assert rc == 1 and "OSTRZEŻENIE" in text          # 4 samples = 585 m of path on a 0.5 m baseline
    geo = m.bearing_from_lag(0)
    assert abs(geo["theta_deg"] - 90.0) < 1e-9


# ------------------------------------------------------------------ checkIfJamming
def test_check_if_jamming(backend, golden_dir, g1_raw, tmp_path):
    import checkIfJamming as m
    g = np.load(os.path.join(golden_dir, "g1_power.npz"))
    path = write_files(tmp_path, [g1_raw], "cij")[0]
    events = m.analyze_file_for_jamming(path, float(g["cij_threshold"]))
    assert [list(e) for e in events] == g["cij_events"].tolist()
    hot, pw = m.analyze_chunk_power(g1_raw[:131072], float(g["cij_threshold"]))
    assert hot is False
    np.testing.assert_allclose(pw, g["cij_power"][0], rtol=1e-6)
    assert m.analyze_chunk_power(g1_raw[:131071], 0.0) == (False, 0.0)
    out = io.StringIO()
    with redirect_stdout(out):
        m.calibrate_file(path)
    import re
    hit = re.search(r"Sugerowany <próg_mocy> \(Mediana \* 4.8\): ([\d.]+)", out.getvalue())
    assert hit and abs(float(hit.group(1)) - float(g["cij_threshold"])) < 0.01


# ------------------------------------------------------------------ worker.GPSAnalysisThread
def collect(th, name):
    got = []
    getattr(th, name).connect(lambda *a: got.append(a))
    return got


def test_worker_power_scan(backend, golden_dir, golden_meta, g1_raw, tmp_path):
    import worker
    g = np.load(os.path.join(golden_dir, "g1_power.npz"))
    path = write_files(tmp_path, [g1_raw], "scan")[0]
    th = quiet(worker.GPSAnalysisThread, [path], power_threshold=9.0)
    assert th.total_file_bytes == int(g["total_file_bytes"]) and th.total_samples == int(g["total_samples"])
    assert th.jamming_byte_ranges == [] and th.power_map_ready is False
    progress = collect(th, "progress_update")
    quiet(th.precalculate_power_profile)
    assert th.power_map_ready is True
    assert isinstance(th.power_map, np.ndarray) and th.power_map.dtype == np.float32
    np.testing.assert_allclose(th.power_map, g["power_map"], rtol=1e-6)
    assert type(th.global_baseline_power).__name__ == golden_meta["g1"]["baseline_type"] == "float32"
    np.testing.assert_allclose(th.global_baseline_power, g["baseline"], rtol=1e-6)
    assert [list(map(int, r)) for r in th.jamming_byte_ranges] == g["ranges"].tolist()
    assert progress[0] == (0, "scanning_power") and progress[-1] == (10, "scanning_power_done")
    assert all(p[0] % 2 == 0 and p[1].startswith("scanning_power") for p in progress)
    # a missing file leaves the state untouched, as in the reference
    th2 = quiet(worker.GPSAnalysisThread, [str(tmp_path / "none.bin")])
    quiet(th2.precalculate_power_profile)
    assert th2.power_map_ready is False


def test_worker_detector_replay(golden_dir):
    """Telemetry state machine against the reference's emitted signal sequence (G5)."""
    import worker
    from golden_inputs import g5_records, g5_scenario
    want = json.load(open(os.path.join(golden_dir, "g5_replay_expected.json")))
    reduced = json.load(open(os.path.join(golden_dir, "g5_capture1_reduced.json")))
    records = g5_records(reduced)
    assert len(records) == want["n_records"]
    th = quiet(worker.GPSAnalysisThread, [])
    sc = g5_scenario()
    th.power_map, th.global_baseline_power, th.jamming_byte_ranges = sc["power_map"], sc["baseline"], sc["ranges"]
    th.power_map_ready, th.total_file_bytes = True, sc["total_file_bytes"]
    th.total_samples = th.estimated_total_samples = sc["total_file_bytes"] // 2
    text, pos, rt, prog = (collect(th, n) for n in ("new_analysis_text", "new_position_data",
                                                    "jamming_detected_realtime", "progress_update"))
    with redirect_stdout(io.StringIO()):
        for r in records:
            th.process_incoming_data(r)
    assert [t[0] for t in text] == want["new_analysis_text"]
    assert [list(map(float, p)) for p in pos] == want["new_position_data"]
    assert [[bool(a), json.loads(json.dumps(b, default=float))] for a, b in rt] == want["jamming_detected_realtime"]
    assert [[int(a), b] for a, b in prog] == want["progress_update"]
    assert json.loads(json.dumps(th.jamming_events, default=float)) == want["jamming_events"]
    assert json.loads(json.dumps(th.last_position_before_jamming, default=float)) == want["last_position_before_jamming"]
    assert bool(th.jamming_detected) == want["final"]["jamming_detected"]
    assert float(th.median_cn0) == want["final"]["median_cn0"]
    assert float(th.current_iq_power) == want["final"]["current_iq_power"]
    res = th.build_result_list()
    assert res[0]["type"] == "jamming" and res[0]["event_number"] == 1 and res[0]["triangulation"] is None
    assert set(res[0]) == {"type", "event_number", "start_sample", "end_sample", "start_time", "end_time",
                           "duration", "triangulation"}
    assert quiet(worker.GPSAnalysisThread, []).build_result_list() == [{"type": "no_jamming"}]


def test_worker_run_end_to_end(backend, g3_raws, tmp_path, monkeypatch):
    """run(): HTTP receiver + scan + (missing) gnssdec + triangulation at end of file while an
    event is open + analysis_complete, through the thread object the GUI would hold."""
    import socket
    import worker
    s = socket.socket()
    try:
        s.bind(("127.0.0.1", 1234))
    except OSError:
        pytest.skip("port 1234 busy")
    finally:
        s.close()
    paths = write_files(tmp_path, g3_raws, "test")          # test0.. -> also picked as test files? names differ
    th = quiet(worker.GPSAnalysisThread, paths, antenna_positions={"antenna1": [0.0, 0.0], "antenna2": [0.5, 0.0],
                                                                  "antenna3": [0.0, 0.5]})
    done, tri, prog = collect(th, "analysis_complete"), collect(th, "triangulation_complete"), collect(th, "progress_update")

    def fake_gnssdec(cmd, **kw):
        # stands in for the subprocess: telemetry arrives over HTTP while it "runs"
        import urllib.request
        assert cmd[1] == "-g" and cmd[-1] == paths[0]
        for i, buff in enumerate((1000, 120000, 131072 * 2, 290000)):
            rec = {"elapsed_time": 0.1 * i, "position": {"nsat": 5, "lat": 50.01, "lon": 19.9, "hgt": 200.0,
                                                          "gdop": 2.0, "clk_bias": 0.0, "buffcnt": buff},
                   "observations": [{"snr": 45.0, "residual": 1.0}]}
            req = urllib.request.Request("http://127.0.0.1:1234/data", data=json.dumps(rec).encode(),
                                         headers={"Content-Type": "application/json"})
            assert urllib.request.urlopen(req, timeout=10).read() == b'{"status":"ok"}'
        raise FileNotFoundError("gnssdec is not part of this package")

    monkeypatch.setattr(worker.subprocess, "run", fake_gnssdec)
    with redirect_stdout(io.StringIO()):
        th.start()
        assert th.wait(60000)
    th.shutdown_server()
    th.shutdown_server()                                    # idempotent (ui_mainwindow.py:825-826)
    assert th.power_map_ready and len(th.power_map) == 5
    assert (100, "completed") in prog
    assert len(done) == 1
    result = done[0][0]
    assert result[0]["type"] == "jamming" and result[0]["start_sample"] == th.jamming_byte_ranges[0][0]
    assert len(tri) == 1 and tri[0][0]["success"] is True
    assert result[0]["triangulation"] is tri[0][0] is th.get_triangulation_result()
    assert tri[0][0]["reference_position"]["valid"] is True


# ------------------------------------------------------------------ widmo_plot
@pytest.mark.gpu
@pytest.mark.parametrize("nperseg", [1024, 4096])
def test_widmo_waterfall(golden_dir, g2_raw, tmp_path, monkeypatch, nperseg):
    import widmo_plot as m
    from oracle import gpsjam_oracle as orc
    monkeypatch.setattr(gpsjam, "_default", None)
    g = np.load(os.path.join(golden_dir, "g2_welch.npz"))
    path = write_files(tmp_path, [g2_raw], "widmo")[0]
    res = quiet(m.analyze_full_file, path, fft_size=nperseg)
    want_db = g[f"db_{nperseg}"]
    assert res["spectrogram"].shape == want_db.shape and res["spectrogram"].dtype == np.float32
    np.testing.assert_allclose(res["spectrogram"], want_db, atol=5e-4)
    np.testing.assert_allclose(res["mean_spectrum"], want_db.mean(axis=0), atol=5e-4)
    _, _, hist_samples = orc.widmo_waterfall(g2_raw, nperseg=nperseg)
    np.testing.assert_array_equal(res["histogram"], np.bincount(hist_samples, minlength=256))
    assert abs(res["duration_sec"] - g2_raw.size / 2 / 2.048e6) < 1e-9
    assert res["freq_axis_mhz"].shape == (nperseg,)
