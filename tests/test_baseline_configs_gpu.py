"""BASELINE.json configs[0..3] at their stated sizes, through the drop-in modules / the C-ABI,
against the CPU oracle on the same bytes (configs[1] at 1 GiB is test_full_size_1gib_properties
in test_gpu_parity.py; configs[4] is the multi-GPU bench, rehearsed in test_sharded_*).

  configs[0]  one 10-s capture (40 960 000 B) through GPSAnalysisThread's power scan
  configs[2]  3 antennas x 10 s: per-stream amplitude statistics -> triangulate_jammer_location
  configs[3]  3 antennas, 2^19-sample slices, 2^20-point cross-correlation, 3 pairs, lags bit-exact
"""
import io
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

import gpsjam                                    # noqa: E402
from gpsjam.synth import StreamSpec, generate    # noqa: E402
from oracle import gpsjam_oracle as orc          # noqa: E402

pytestmark = pytest.mark.gpu

TEN_SECONDS = 20_480_000          # samples = 40 960 000 bytes at 2.048 MS/s
POSITIONS = [[0.0, 0.0], [0.5, 0.0], [0.0, 0.5]]
GAINS = (1.0, 0.7, 0.5)


def _capture(antenna, delay=0, n=TEN_SECONDS):
    spec = StreamSpec(seed=1234 + antenna, antenna=antenna, delay=delay, jam_start=int(0.4 * n), jam_end=int(0.7 * n),
                      noise_sigma=6.25, jam_sigma=40.0 * GAINS[antenna])
    return generate(spec, n)


def _quiet(fn, *a, **k):
    with redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def test_config0_worker_power_scan(tmp_path, monkeypatch):
    monkeypatch.setattr(gpsjam, "_default", None)
    import worker
    raw = _capture(0)
    path = tmp_path / "capture.bin"
    raw.tofile(path)
    th = worker.GPSAnalysisThread([str(path)], power_threshold=6.0)
    _quiet(th.precalculate_power_profile)
    pm = orc.chunk_power(raw)
    base, thr, ranges = orc.power_threshold(pm)
    assert th.power_map_ready and th.total_file_bytes == raw.size and th.power_map.shape == (625,)
    np.testing.assert_allclose(th.power_map, pm, rtol=1e-6)
    assert np.float32(th.global_baseline_power) == np.float32(base)
    assert [(int(a), int(b)) for a, b in th.jamming_byte_ranges] == [(int(a), int(b)) for a, b in ranges]
    assert len(ranges) == 1      # one contiguous jamming interval (edges on chunk boundaries)


def test_config2_three_antenna_rssi(tmp_path, monkeypatch):
    monkeypatch.setattr(gpsjam, "_default", None)
    import triangulateRSSI as m
    raws = [_capture(a) for a in range(3)]
    paths = []
    for k, r in enumerate(raws):
        p = tmp_path / f"ant{k}.bin"
        r.tofile(p)
        paths.append(str(p))
    for thr in (0.0, 0.1):
        got = _quiet(m.triangulate_jammer_location, paths, antenna_positions_meters=[np.array(p) for p in POSITIONS],
                     threshold=thr, verbose=False)
        want = orc.triangulate(raws, antenna_positions_meters=[np.array(p) for p in POSITIONS], threshold=thr)
        assert got["success"] is True and want["success"] is True
        np.testing.assert_allclose(got["distances"], want["distances"], rtol=1e-5)
        # the grid spans 1.5 max(distance) around the antenna centroid, so its points move with the
        # distances (1e-7 apart: float32 pairwise mean over 2 x 10^7 amplitudes in the reference vs
        # float64 accumulation here): same grid node, coordinates equal to that precision
        np.testing.assert_allclose(got["location_meters"], want["location_meters"], rtol=1e-6)

        def node(res):
            half = 1.5 * max(res["distances"])
            centre = np.mean(np.array(POSITIONS), axis=0)
            return tuple(int(round((res["location_meters"][k] - (centre[k] - half)) / (2 * half / 299))) for k in (0, 1))
        assert node(got) == node(want)
        for k, v in want["location_geographic"].items():
            np.testing.assert_allclose(got["location_geographic"][k], v, rtol=1e-6)


def test_config3_tdoa_three_pairs(dev):
    n = 1 << 19
    delays = (0, 3, -5)
    nsamp = 1_000_000 + n + 64
    raws = [generate(StreamSpec(seed=1234, antenna=a, delay=d, jam_start=1_000_000, jam_end=1 << 40, noise_sigma=6.25,
                                jam_sigma=40.0), nsamp) for a, d in enumerate(delays)]
    onset = [dev.onset(r, 200000, 1000, 50.0).start_index for r in raws]
    zs = [orc.tdoa_unpack(r) for r in raws]
    assert onset == [orc.tdoa_onset(z) for z in zs]
    start = min(onset)                                   # a common window, so lags are the true delays
    slices = [r[2 * start:2 * (start + n)] for r in raws]
    pairs = [(0, 1), (0, 2), (1, 2)]
    lags, peaks = dev.xcorr_lags(slices, pairs)
    for (a, b), lag, pk in zip(pairs, lags, peaks):
        want, wpk = orc.xcorr_lag(zs[b][start:start + n], zs[a][start:start + n])
        assert lag == want == delays[b] - delays[a]
        np.testing.assert_allclose(pk, wpk, rtol=1e-4)
