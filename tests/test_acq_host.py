"""Host side of the acquisition search (gpsjam/gnss.py) on the CPU: the C/A codes against the
published IS-GPS-200 known answers (the one part of this row that IS pinned -- the reference's
acquisition lives in gnssdec, which cannot be built here: parity unpinned for the FFT stages),
and the product's code / resampling / Doppler-bin / mixer-phase tables against the oracle's
line-by-line restatement of the C."""
import numpy as np
import pytest

from gpsjam import gnss
from oracle import gpsjam_oracle as orc

# IS-GPS-200 Table 3-Ia, "first 10 chips (octal)" of the C/A code, PRN 1..32
FIRST_10_CHIPS_OCTAL = (1440, 1620, 1710, 1744, 1133, 1455, 1131, 1454, 1626, 1504, 1642, 1750, 1764, 1772, 1775,
                        1776, 1156, 1467, 1633, 1715, 1746, 1763, 1063, 1706, 1743, 1761, 1770, 1774, 1127, 1453,
                        1625, 1712)


@pytest.mark.parametrize("prn", range(1, 33))
def test_ca_code_matches_is_gps_200(prn):
    want = int(str(FIRST_10_CHIPS_OCTAL[prn - 1]), 8)
    for code in (gnss.ca_code(prn), orc.acq_gencode_l1ca(prn)):
        assert code.dtype == np.int16 and code.size == 1023 and set(np.unique(code)) == {-1, 1}
        bits = (code[:10] > 0).astype(int)                     # +1 = logical one
        assert int("".join(map(str, bits)), 2) == want
        assert abs(int(code.sum())) == 1                       # balance property of a Gold code: 512 vs 511
    np.testing.assert_array_equal(gnss.ca_code(prn), orc.acq_gencode_l1ca(prn))


def test_code_autocorrelation_is_three_valued():
    c = gnss.ca_code(7).astype(np.int64)
    r = np.array([np.dot(c, np.roll(c, k)) for k in range(1023)])
    assert r[0] == 1023 and set(np.unique(r[1:])) <= {-65, -1, 63}


@pytest.mark.parametrize("nsamp,fs", [(2048, 2.048e6), (1024, 1.024e6), (2048, 2.048e6 * (1 + 1e-9))])
def test_resampling_matches_oracle(nsamp, fs):
    for prn in (1, 13, 32):
        code = gnss.ca_code(prn)
        got = gnss.resample_code(code, nsamp, fs)
        want = orc.acq_rescode(code, (1.0 / fs) * 1.023e6, nsamp)
        np.testing.assert_array_equal(got, want)
    # at 2.048 MHz the phase arithmetic is exact: sample n carries chip floor(n 1023 / 2048)
    n = np.arange(2048)
    np.testing.assert_array_equal(gnss.resample_code(gnss.ca_code(5), 2048), gnss.ca_code(5)[(n * 1023) // 2048])


def test_doppler_bins():
    f = gnss.doppler_bins()
    assert f.size == 71 and f[0] == -7000.0 and f[35] == 0.0 and f[-1] == 7000.0 and np.all(np.diff(f) == 200.0)


def test_mixer_phase_table_matches_oracle_mixer():
    """The table must reproduce the reference mixer's outputs exactly, negative Doppler included
    (cvttpd truncates toward zero, so a phase of -0.3 gives index 0 but -1.2 gives -1 & 15 = 15)."""
    rng = np.random.default_rng(2)
    m = 4096
    data = rng.integers(-128, 128, 2 * m).astype(np.int8)
    freqs = gnss.doppler_bins()
    tab = gnss.mixer_phase_table(freqs, 2.048e6, m)
    assert tab.shape == (71, m) and tab.max() <= 15
    cos = np.array([32, 30, 23, 12, 0, -12, -23, -30, -32, -30, -23, -12, 0, 12, 23, 30])
    sin = np.array([0, 12, 23, 30, 32, 30, 23, 12, 0, -12, -23, -30, -32, -30, -23, -12])
    di, dq = data[0::2].astype(np.int32), data[1::2].astype(np.int32)
    for k in (0, 1, 17, 34, 35, 36, 52, 70):
        II, QQ = orc.acq_mixcarr_sse2(data, 1 / 2.048e6, m, float(freqs[k]))
        np.testing.assert_array_equal(cos[tab[k]] * di - sin[tab[k]] * dq, II)
        np.testing.assert_array_equal(sin[tab[k]] * di + cos[tab[k]] * dq, QQ)
    assert (tab[35] == 0).all()                              # zero Doppler: the phase never moves
    # -7 kHz: 0.0547 of a table step per sample, truncated toward zero: samples 0..18 stay at 0, then 15, 14 ...
    assert tab[0, 18] == 0 and tab[0, 19] == 15 and tab[0, 37] == 14
    assert tab[70, 18] == 0 and tab[70, 19] == 1


def test_pcorrelator_restatement_against_the_definition():
    """What pcorrelator + cpxconv compute, read off the C (sdrcmn.c:124-147,742-773) without any FFT:
    P[k] = (CSCALE / m)^2 |sum_n (II + j QQ)[(n + k) mod m] code[n]|^2 -- exact integer arithmetic here.
    The oracle's FFT form (complex64, like FFTW's) must agree to float32 accuracy: this pins the conjugation,
    the sign and the m^2 normalisation of the restatement, though not the reference's own rounding."""
    rng = np.random.default_rng(11)
    nsamp, m = 2048, 4096
    data = rng.integers(-40, 41, 2 * m).astype(np.int8)
    freqs = [-3400.0, 0.0, 1200.0]
    codex = orc.acq_code_fft(9, nsamp)
    P = np.zeros(len(freqs) * nsamp)
    orc.acq_pcorrelator(data, 1 / 2.048e6, nsamp, freqs, m, codex, P)
    rcode = np.zeros(m, np.int64)
    rcode[:nsamp] = orc.acq_rescode(orc.acq_gencode_l1ca(9), (1 / 2.048e6) * 1.023e6, nsamp)
    for i, f in enumerate(freqs):
        II, QQ = orc.acq_mixcarr_sse2(data, 1 / 2.048e6, m, f)
        II, QQ = II.astype(np.int64), QQ.astype(np.int64)
        want = np.empty(nsamp)
        for k in range(0, nsamp, 97):                       # every 97th code phase: exact, a few dozen dot products
            re = int(np.dot(np.roll(II, -k), rcode))
            im = int(np.dot(np.roll(QQ, -k), rcode))
            want[k] = (re * re + im * im) * (orc.ACQ_CSCALE / m) ** 2
        got = P[i * nsamp:(i + 1) * nsamp]
        ks = np.arange(0, nsamp, 97)
        np.testing.assert_allclose(got[ks], want[ks], rtol=0, atol=2e-5 * want[ks].max())
