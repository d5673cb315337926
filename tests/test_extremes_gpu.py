"""Inputs at the edge of the uint8 range -- what a receiver next to a strong jammer records (a saturated ADC: runs of
0 and 255), constant bytes, the Nyquist pattern at full scale, uniform bytes over the whole range, one impulse -- through
K1-K5 against the oracle.  Tolerances as everywhere (module docstring of test_gpu_parity.py)."""
import numpy as np
import pytest

from oracle import gpsjam_oracle as orc

pytestmark = pytest.mark.gpu

N = 300000   # samples


def rel_err(got, want, floor=1e-12):
    keep = want > floor
    return float(np.max(np.abs(got[keep] - want[keep]) / want[keep])) if keep.any() else 0.0


def _interleave(i, q):
    raw = np.empty(2 * i.size, np.uint8)
    raw[0::2], raw[1::2] = i, q
    return raw


def extreme_captures():
    rng = np.random.RandomState(11)
    t = np.arange(N)
    out = {}
    out["constant 255"] = np.full(2 * N, 255, np.uint8)
    out["constant 0"] = np.zeros(2 * N, np.uint8)
    out["constant 128/127"] = _interleave(np.full(N, 128, np.uint8), np.full(N, 127, np.uint8))
    out["nyquist full scale"] = _interleave(np.where(t & 1, 255, 0).astype(np.uint8), np.where(t & 1, 0, 255).astype(np.uint8))
    out["uniform bytes"] = rng.randint(0, 256, 2 * N).astype(np.uint8)
    sat = 400.0 * np.exp(2j * np.pi * 0.031 * t) + rng.normal(0, 30, N) + 1j * rng.normal(0, 30, N)   # clips hard
    out["saturated tone"] = _interleave((np.clip(np.rint(sat.real), -128, 127) + 128).astype(np.uint8),
                                        (np.clip(np.rint(sat.imag), -128, 127) + 128).astype(np.uint8))
    quiet = (np.clip(np.rint(rng.normal(0, 2.0, 2 * N)), -128, 127) + 128).astype(np.uint8)
    burst = quiet.copy()
    burst[2 * 220000:] = np.where(rng.randint(0, 2, 2 * (N - 220000)) == 1, 255, 0)        # rail to rail after 220000
    out["quiet then rail to rail"] = burst
    imp = np.full(2 * N, 128, np.uint8)
    imp[2 * 123457] = 255
    out["one impulse"] = imp
    return out


CAPTURES = extreme_captures()


@pytest.mark.parametrize("name", list(CAPTURES))
def test_extreme_k1_power_map(dev, name):
    raw = CAPTURES[name]
    for chunk_bytes in (65536, 131072):
        pm = dev.chunk_power(raw, chunk_bytes=chunk_bytes)
        np.testing.assert_allclose(pm, orc.chunk_power(raw, chunk_bytes), rtol=1e-6)


@pytest.mark.parametrize("name", list(CAPTURES))
@pytest.mark.parametrize("nperseg", [256, 4096])
def test_extreme_k2_welch(dev, name, nperseg):
    raw = CAPTURES[name]
    psd, _ = dev.welch(raw, chunk_samples=100000, nperseg=nperseg, want_db=False)
    lin, _, _ = orc.widmo_waterfall(raw, nperseg=nperseg, chunk_samples=100000)
    assert psd.shape == lin.shape and np.all(np.isfinite(psd)) and np.all(psd >= 0)
    # relative to each row's own peak for the bins the reference leaves at rounding-noise level (a constant capture is
    # all such bins: scipy's detrend leaves ~1e-17, the frequency-domain detrend ~1e-13 of a full-scale bin)
    scale = lin.max(axis=1, keepdims=True)
    big = lin > 1e-6 * np.maximum(scale, 1e-30)
    if big.any():
        assert float(np.max(np.abs(psd[big] - lin[big]) / lin[big])) < 1e-4, name
    full_scale = 1.0 / (2.048e6 * 0.375 * nperseg) * nperseg ** 2 * 2.0           # PSD of a full-scale tone, one bin
    # every bin, the rounding-noise ones included: off by less than 2e-5 of the row's peak (1e-12 of full scale for a
    # capture whose true spectrum is empty)
    assert float(np.max(np.abs(psd - lin))) < 2e-5 * max(float(scale.max()), 5e-8 * full_scale), name


@pytest.mark.parametrize("name", list(CAPTURES))
def test_extreme_k3_amp_stats(dev, name):
    raw = CAPTURES[name]
    for thr in (0.0, 0.5, 1.2, 1.5):
        st = dev.amp_stats(raw, thr)
        k, avg = orc.rssi_amp_stats(raw, thr)
        if k is None:
            assert st.first_index == -1 and st.count == 0, (name, thr)
        else:
            assert st.first_index == k and st.count == N - k, (name, thr)
            np.testing.assert_allclose(st.mean, avg, rtol=1e-6)


@pytest.mark.parametrize("name", list(CAPTURES))
def test_extreme_k4_onset(dev, name):
    raw = CAPTURES[name]
    z = orc.tdoa_unpack(raw)
    for noise, window, factor in ((200000, 1000, 50.0), (1000, 64, 3.0)):
        got = dev.onset(raw, noise, window, factor)
        assert got.start_index == orc.tdoa_onset(z, noise, window, factor), (name, noise, window, got.margin)


@pytest.mark.parametrize("name", list(CAPTURES))
def test_extreme_k5_lags(dev, name):
    raw = CAPTURES[name]
    n = 50000
    a = raw[2 * 200000:2 * (200000 + n)]
    b = raw[2 * (200000 - 17):2 * (200000 - 17 + n)]          # the same signal 17 samples later
    lags, peaks, margins = dev.xcorr_lags([a, b], [(0, 1), (1, 0)], want_margins=True)
    want01, pk01 = orc.xcorr_lag(orc.tdoa_unpack(b), orc.tdoa_unpack(a))
    want10, _ = orc.xcorr_lag(orc.tdoa_unpack(a), orc.tdoa_unpack(b))
    # constant / periodic captures: the correlation is a triangle, the runner-up sits (N-1)/N below the peak
    assert lags[0] == want01 and lags[1] == want10, (name, lags, margins)
    if margins[0] > 1e-3:
        assert want01 == 17 and want10 == -17
        np.testing.assert_allclose(peaks[0], pk01, rtol=1e-4)
    else:
        np.testing.assert_allclose(margins, 1.0 / n, rtol=0.02)


@pytest.mark.parametrize("name", list(CAPTURES))
def test_extreme_fused_scan(dev, name):
    """The one-pass scan of the pipeline (gj_stream_scan_dev) on the same captures: power map, amplitude statistics and
    onset as from K1 / K3 / K4 alone and as the oracle has them."""
    raw = CAPTURES[name]
    nbytes, chunk = raw.size, 65536
    buf = dev.alloc(nbytes).upload(raw)
    nch = dev.chunk_count(nbytes, chunk)
    z = orc.tdoa_unpack(raw)
    for thr, (noise, window, factor) in ((0.0, (200000, 1000, 50.0)), (1.2, (1000, 64, 3.0))):
        d_pow, d_amp, d_on = dev.alloc(4 * nch), dev.alloc(32), dev.alloc(32)
        dev.stream_scan_dev(buf, nbytes, chunk, d_pow, thr, d_amp, noise, window, factor, d_on)
        dev.synchronize()
        amp = np.frombuffer(d_amp.download(np.uint8).tobytes(), dtype=[("i", "<i8"), ("c", "<u8"), ("s", "<f8"),
                                                                       ("m", "<f4"), ("r", "<f4")])[0]
        np.testing.assert_allclose(d_pow.download(np.float32, nch), orc.chunk_power(raw, chunk), rtol=1e-6)
        k, avg = orc.rssi_amp_stats(raw, thr)
        if k is None:
            assert amp["i"] == -1 and amp["c"] == 0, (name, thr)
        else:
            assert amp["i"] == k and amp["c"] == N - k, (name, thr)
            np.testing.assert_allclose(amp["m"], avg, rtol=1e-6)
        assert int(d_on.download(np.int64, 1)[0]) == orc.tdoa_onset(z, noise, window, factor), (name, noise, window)
        for b in (d_pow, d_amp, d_on):
            b.free()
    buf.free()
