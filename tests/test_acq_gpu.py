"""SURVEY 8(f)-4 on the GPU: the batched acquisition search (gj_acq_search_dev) against the oracle's
line-by-line restatement of the reference receiver's per-channel search (sdracq.c / sdrcmn.c).
PARITY UNPINNED behind the first FFT -- gnssdec cannot be built here, see oracle section (6); the
integer mixer and the code tables are pinned (tests/test_acq_host.py).  Tolerances: correlation
power 2e-4 of the row's peak (two different complex64 FFTs), indices and step counts exact."""
import numpy as np
import pytest

import gpsjam
from gpsjam import gnss
from oracle import gpsjam_oracle as orc

pytestmark = pytest.mark.gpu

FS = 2.048e6


def gps_like_capture(n, sats, noise_sigma=12.0, seed=4):
    """uint8 I/Q: sum of C/A signals (prn, doppler_hz, code_delay_samples, amplitude) + Gaussian noise, +128."""
    rng = np.random.default_rng(seed)
    t = np.arange(n) / FS
    z = rng.normal(0, noise_sigma, n) + 1j * rng.normal(0, noise_sigma, n)
    for prn, dop, delay, amp in sats:
        code = gnss.ca_code(prn).astype(np.float64)
        chip = ((np.arange(n) - delay) * 1.023e6 / FS) % 1023
        z += amp * code[chip.astype(np.int64)] * np.exp(-2j * np.pi * dop * t)   # the reference mixer wipes off exp(-j 2 pi f t)
    iq = np.empty(2 * n, np.float64)
    iq[0::2], iq[1::2] = z.real, z.imag
    return (np.clip(np.round(iq), -128, 127) + 128).astype(np.uint8)


def test_acq_search_matches_oracle(dev):
    sats = [(3, 1400.0, 517, 9.0), (17, -3000.0, 1201, 1.5), (25, 5230.0, 88, 3.0)]
    n = 14 * 2048
    raw = gps_like_capture(n, sats)
    prns = [3, 8, 17, 25]
    first = 1000
    srch = gnss.AcqSearch(dev, prns=prns)
    assert srch.nsamp == 2048 and srch.nsampchip == 2 and len(srch.freqs) == 71 and srch.samples_needed() == 11 * 2048
    with dev.capture(raw) as cap:
        res, P = srch.search(cap, first_sample=first, want_power=True)
        again = srch.search(cap, first_sample=first)                 # second run: workspace state is reset
    # the single-launch form (row records, P in registers) replays the same decisions as the step-by-step form
    assert [(r.acquired, r.code_index, r.freq_index, r.steps) for r in res] == \
           [(r.acquired, r.code_index, r.freq_index, r.steps) for r in again]
    for a, b in zip(res, again):
        np.testing.assert_allclose([a.max_power, a.second_power, a.mean_power, a.peak_ratio, a.cn0],
                                   [b.max_power, b.second_power, b.mean_power, b.peak_ratio, b.cn0], rtol=1e-12)
    by_prn = {r.prn: r for r in res}
    for k, prn in enumerate(prns):
        want, Pw = orc.acq_search(raw, first, prn)
        got = by_prn[prn]
        assert got.acquired == want["acquired"], (prn, got, want)
        assert got.steps == want["steps"], (prn, got.steps, want["steps"])
        Pw = Pw.reshape(71, 2048)
        np.testing.assert_allclose(P[k], Pw, rtol=0, atol=2e-4 * Pw.max())
        if want["peakr"] > 1.5:                                    # an unambiguous peak: same cell
            assert (got.code_index, got.freq_index) == (want["codei"], want["freqi"]), (prn, got, want)
            np.testing.assert_allclose([got.max_power, got.second_power, got.mean_power, got.peak_ratio],
                                       [want["maxP"], want["maxP2"], want["meanP"], want["peakr"]], rtol=2e-3)
            np.testing.assert_allclose(got.cn0, want["cn0"], atol=0.02)
    # the three satellites in the capture are found where they were put, the absent PRN is not
    for prn, dop, delay, _ in sats:
        r = by_prn[prn]
        assert r.acquired and abs(r.doppler_hz - dop) <= 200.0
        assert min(abs(r.code_index - (delay - first) % 2048), 2048 - abs(r.code_index - (delay - first) % 2048)) <= 1
    assert not by_prn[8].acquired and by_prn[8].steps == 10
    assert by_prn[3].steps == 1 and by_prn[17].steps > 1             # the weak one needs integration
    srch.close()


def test_acq_all_32_prns_and_errors(dev):
    sats = [(p, 200.0 * ((7 * p) % 60 - 30), (97 * p) % 2048, 6.0) for p in (1, 9, 14, 22, 31)]
    raw = gps_like_capture(12 * 2048, sats, noise_sigma=10.0, seed=9)
    srch = gnss.AcqSearch(dev)
    with dev.capture(raw) as cap:
        res = srch.search(cap)
        assert sorted(r.prn for r in res if r.acquired) == [1, 9, 14, 22, 31]
        for prn, dop, delay, _ in sats:
            r = res[prn - 1]
            assert abs(r.doppler_hz - dop) <= 200.0 and abs(r.code_index - delay) <= 1
        with pytest.raises(gpsjam.GpsJamError):
            srch.search(cap, first_sample=2 * 2048)                  # the eleven milliseconds do not fit any more
        with pytest.raises(gpsjam.GpsJamError):
            srch.search(cap, first_sample=(1 << 64) - 4096)          # would wrap the bounds arithmetic
    srch.close()


@pytest.mark.parametrize("fs", [1.024e6, 0.512e6])
def test_acq_other_sampling_rates(dev, fs):
    """nsamp = 1024 / 512 (FFT length 2048 / 1024): two / four transforms per workgroup, other
    reduction shapes.  Same checks against the oracle as at 2.048 MS/s."""
    nsamp = int(fs * 1e-3)
    rng = np.random.default_rng(int(fs) % 97)
    n = 13 * nsamp
    t = np.arange(n) / fs
    z = rng.normal(0, 10.0, n) + 1j * rng.normal(0, 10.0, n)
    sats = [(4, 1800.0, nsamp // 3, 7.0), (20, -4400.0, nsamp - 9, 2.0)]
    for prn, dop, delay, amp in sats:
        chip = ((np.arange(n) - delay) * 1.023e6 / fs) % 1023
        z += amp * gnss.ca_code(prn)[chip.astype(np.int64)] * np.exp(-2j * np.pi * dop * t)
    iq = np.empty(2 * n)
    iq[0::2], iq[1::2] = z.real, z.imag
    raw = (np.clip(np.round(iq), -128, 127) + 128).astype(np.uint8)
    prns = [4, 9, 20]
    srch = gnss.AcqSearch(dev, prns=prns, fs=fs)
    assert srch.nsamp == nsamp
    with dev.capture(raw) as cap:
        res, P = srch.search(cap, first_sample=17, want_power=True)
        fast = srch.search(cap, first_sample=17)
    for k, prn in enumerate(prns):
        want, Pw = orc.acq_search(raw, 17, prn, fs=fs)
        Pw = Pw.reshape(71, nsamp)
        np.testing.assert_allclose(P[k], Pw, rtol=0, atol=2e-4 * Pw.max())
        for got in (res[k], fast[k]):
            assert got.acquired == want["acquired"] and got.steps == want["steps"], (prn, got, want)
            if want["peakr"] > 1.5:
                assert (got.code_index, got.freq_index) == (want["codei"], want["freqi"]), (prn, got, want)
                np.testing.assert_allclose(got.peak_ratio, want["peakr"], rtol=2e-3)
    assert res[0].acquired and abs(res[0].doppler_hz - 1800.0) <= 200.0
    srch.close()
