"""The integer-exact restatements of K1 / K3 / K4 (tests/exact_restatement.py) against the vectors captured from the
reference and against the oracle: they are what the GPU tests compare the kernels with bit for bit, so they are pinned
here, on the CPU, first."""
import os

import numpy as np
import pytest

import exact_restatement as ex
import golden_inputs as gi
from gpsjam.synth import StreamSpec, generate
from oracle import gpsjam_oracle as orc


def test_chunk_power_is_the_references_map(golden_dir, g1_raw):
    g = np.load(os.path.join(golden_dir, "g1_power.npz"))
    np.testing.assert_allclose(ex.chunk_power(g1_raw, 65536), g["power_map"], rtol=1e-6)
    cij = ex.chunk_power(g1_raw, orc.CIJ_CHUNK_BYTES, eps=0.0, odd_chunk_zero=True)
    assert cij[-1] == 0.0                                        # ragged odd tail: checkIfJamming.py:52-55
    np.testing.assert_allclose(cij, g["cij_power"], rtol=1e-6)
    assert np.isnan(ex.chunk_power(np.zeros(65537, np.uint8), 65536)[-1])


def test_onset_is_the_references_index(golden_meta, g4_raws):
    g4 = golden_meta["g4"]
    got = [ex.onset(r) for r in g4_raws]
    assert [o["start"] for o in got] == g4["onset"]
    assert all(o["guard"] == o["start"] and o["hit"] >= 1e-6 for o in got)       # "the reference's by construction"
    assert ex.onset(g4_raws[0][:2 * 200500])["start"] == g4["onset_short"] == -1
    assert ex.onset(g4_raws[0][:2 * 250000])["start"] == g4["onset_none"] == -1
    assert ex.onset(g4_raws[1], 50000, 256, 20.0)["start"] == g4["onset_alt"]
    for o, r in zip(got, g4_raws):                               # noise and threshold: the reference's to float32 rounding
        p = np.abs(orc.tdoa_unpack(r)) ** 2
        np.testing.assert_allclose(o["noise"], np.mean(p[:200000]), rtol=1e-6)
        np.testing.assert_allclose(o["thr"], np.mean(p[:200000]) * 50.0, rtol=1e-6)


def test_amp_stats_are_the_references(golden_meta, g3_raws):
    for key, (idx, avg) in golden_meta["g3"]["amp_stats"].items():
        k, thr_s = key.split("_")
        got = ex.amp_stats(g3_raws[int(k)], float(thr_s))
        assert got["first"] == (-1 if idx is None else idx)
        if idx is not None:
            np.testing.assert_allclose(got["mean"], avg, rtol=1e-6)


@pytest.mark.parametrize("seed,jam,noise,window", [(1, 300_000, 200_000, 1000), (2, 250_123, 100_000, 513), (3, 1 << 40, 200_000, 1000),
                                                  (4, 260_000, 123_457, 8)])
def test_restatements_agree_with_the_oracle_on_synthetic_streams(seed, jam, noise, window):
    raw = generate(StreamSpec(seed=seed, jam_start=jam, jam_end=1 << 40, jam_sigma=60.0), 420_000)
    z = orc.tdoa_unpack(raw)
    assert ex.onset(raw, noise, window, 50.0)["start"] == orc.tdoa_onset(z, noise, window, 50.0)
    for thr in (0.0, 0.3):
        k, avg = orc.rssi_amp_stats(raw, thr)
        got = ex.amp_stats(raw, thr)
        assert got["first"] == (-1 if k is None else k)
        if k is not None:
            np.testing.assert_allclose(got["mean"], avg, rtol=1e-6)
    np.testing.assert_allclose(ex.chunk_power(raw, 65536), orc.chunk_power(raw), rtol=1e-6)
