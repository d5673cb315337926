"""The CPU oracle against the vectors captured from the reference (no GPU, no reference).

Bit-exact where the oracle repeats the reference's numpy/scipy call sequence on the
same library versions; a tight tolerance otherwise so the suite also passes on a box
whose numpy/scipy build sums in a different order.
"""
import json
import os

import numpy as np
import pytest

from conftest import sha256
from oracle import gpsjam_oracle as orc
import golden_inputs as gi


def test_inputs_reproduce(golden_meta, g1_raw, g2_raw, g3_raws, g4_raws):
    assert sha256(g1_raw) == golden_meta["g1"]["sha256"]
    assert sha256(g2_raw) == golden_meta["g2"]["sha256"]
    assert [sha256(r) for r in g3_raws] == golden_meta["g3"]["sha256"]
    assert [sha256(r) for r in g4_raws] == golden_meta["g4"]["sha256"]


def test_g1_power_scan(golden_dir, g1_raw):
    g = np.load(os.path.join(golden_dir, "g1_power.npz"))
    pm = orc.chunk_power(g1_raw)
    assert pm.dtype == np.float32 and pm.shape == g["power_map"].shape == (21,)
    np.testing.assert_allclose(pm, g["power_map"], rtol=1e-6)
    base, thr, ranges = orc.power_threshold(pm)
    np.testing.assert_allclose(base, g["baseline"], rtol=1e-6)
    assert [list(r) for r in ranges] == g["ranges"].tolist() == [[655360, 917504]]


def test_g1_check_if_jamming(golden_dir, g1_raw):
    g = np.load(os.path.join(golden_dir, "g1_power.npz"))
    pw = [orc.cij_chunk_power(g1_raw[o:o + orc.CIJ_CHUNK_BYTES], 0.0)[1]
          for o in range(0, g1_raw.size, orc.CIJ_CHUNK_BYTES)]
    # the ragged tail is odd-sized -> (False, 0.0)
    assert pw[-1] == 0.0 and g["cij_power"][-1] == 0.0
    np.testing.assert_allclose(np.array(pw, dtype=np.float64), g["cij_power"], rtol=1e-6)
    ev = orc.cij_events(g1_raw, float(g["cij_threshold"]))
    assert [list(e) for e in ev] == g["cij_events"].tolist()
    med, mx, mn, sug = orc.cij_calibrate(g1_raw)
    np.testing.assert_allclose(sug, g["cij_threshold"], rtol=1e-6)


@pytest.mark.parametrize("nperseg", [1024, 4096])
def test_g2_welch(golden_dir, g2_raw, nperseg):
    g = np.load(os.path.join(golden_dir, "g2_welch.npz"))
    lin, db, hist = orc.widmo_waterfall(g2_raw, nperseg=nperseg)
    ref = g[f"lin_{nperseg}"]
    assert lin.shape == ref.shape == (2, nperseg) and lin.dtype == ref.dtype == np.float32
    np.testing.assert_allclose(lin, ref, rtol=2e-5)
    np.testing.assert_allclose(db, g[f"db_{nperseg}"], atol=1e-4)
    assert hist.size == -(-4096000 // 100) + -(-600000 // 100)


def test_g3_rssi(golden_meta, g3_raws):
    g3 = golden_meta["g3"]
    for thr_s, want in g3["distances"].items():
        got = [orc.rssi_distance(r, threshold=float(thr_s)) for r in g3_raws]
        for a, b in zip(got, want):
            if b is None:
                assert a is None
            else:
                np.testing.assert_allclose(a, b, rtol=1e-6)
    for key, (idx, avg) in g3["amp_stats"].items():
        k, thr_s = key.split("_")
        i2, a2 = orc.rssi_amp_stats(g3_raws[int(k)], float(thr_s))
        assert i2 == idx
        np.testing.assert_allclose(a2, avg, rtol=1e-6)
    for thr_s, want in g3["triangulate"].items():
        got = orc.triangulate(g3_raws, [np.array(p) for p in gi.G3_POSITIONS], 50.06, 19.94,
                              threshold=float(thr_s))
        assert got["success"] and want["success"]
        assert got["location_meters"] == want["location_meters"]
        np.testing.assert_allclose(got["distances"], want["distances"], rtol=1e-6)
        assert got["message"] == want["message"] and got["num_antennas"] == want["num_antennas"]
        for k2, v in want["location_geographic"].items():
            np.testing.assert_allclose(got["location_geographic"][k2], v, rtol=1e-12)
    two = orc.triangulate(g3_raws[:2], threshold=0.0)
    assert two["location_meters"] == g3["two_files_default_positions"]["location_meters"]
    one = orc.triangulate(g3_raws[:1])
    assert one == g3["one_file"]
    miss = orc.triangulate([g3_raws[0], None, g3_raws[2]], threshold=0.0)
    assert miss["distances"][1] is None and g3["missing_file"]["distances"][1] is None
    assert miss["location_meters"] == g3["missing_file"]["location_meters"]
    assert miss["num_antennas"] == g3["missing_file"]["num_antennas"] == 2


def test_g4_tdoa(golden_meta, g4_raws):
    g4 = golden_meta["g4"]
    sigs = [orc.tdoa_unpack(r) for r in g4_raws]
    onset = [orc.tdoa_onset(s) for s in sigs]
    assert onset == g4["onset"]
    assert orc.tdoa_onset(sigs[0][:200500]) == g4["onset_short"] == -1
    assert orc.tdoa_onset(sigs[0][:250000]) == g4["onset_none"] == -1
    assert orc.tdoa_onset(sigs[1], 50000, 256, 20.0) == g4["onset_alt"]
    for n in gi.G4_SLICES:
        for a, b in ((0, 1), (0, 2), (1, 2)):
            key = f"{n}_{a}{b}"
            s_a = sigs[a][onset[a]:onset[a] + n]
            lag, peak = orc.xcorr_lag(sigs[b][onset[b]:onset[b] + n], s_a)
            assert lag == g4["lags_own_start"][key]
            np.testing.assert_allclose(peak, g4["peaks"]["own_" + key], rtol=1e-5)
            lag, peak = orc.xcorr_lag(sigs[b][onset[a]:onset[a] + n], s_a)
            assert lag == g4["lags_common_start"][key] == gi.G4_DELAYS[b] - gi.G4_DELAYS[a]


def test_bearing_quirk():
    # lag 0 -> broadside; reference's atan2(dy, 0) baseline angle (triangulateTDOA.py:114)
    r = orc.tdoa_bearing(0, [0, 0], [0.5, 0])
    assert abs(r["theta_deg"] - 90.0) < 1e-12 and abs(r["azimuth1_deg"] - 90.0) < 1e-12
    assert orc.tdoa_bearing(1, [0, 0], [0.5, 0]) is None     # 146 m path > 0.5 m baseline
    assert orc.tdoa_bearing(0, [0, 0], [0, 0]) is None
