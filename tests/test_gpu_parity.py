"""GPU parity: the HIP path (through the C-ABI, via gpsjam.Device) against the CPU oracle and
the golden vectors captured from the reference.  Run with ``-m gpu`` on an MI355X.

Tolerances (north_star / SURVEY.md 8d): power map 1e-6 rel and identical byte ranges; PSD
1e-4 rel on the linear PSD; RSSI amplitude mean 1e-6 rel (distance 1e-5); onset index and
TDOA lag bit-exact.
"""
import os

import numpy as np
import pytest

import gpsjam
from gpsjam.synth import StreamSpec, generate
from oracle import gpsjam_oracle as orc
import exact_restatement as ex
import golden_inputs as gi

pytestmark = pytest.mark.gpu


def exact_chunk_power(raw, chunk_bytes, eps=1e-10):
    """Integer-exact restatement of the chunk power, rounded once (what K1 promises)."""
    out = []
    for off in range(0, raw.size, chunk_bytes):
        piece = raw[off:off + chunk_bytes]
        n = piece.size // 2
        if n == 0:
            out.append(np.float32(np.nan))
            continue
        v = 2 * piece[:2 * n].astype(np.int64) - 255
        s = int(np.sum(v * v))
        out.append(np.float32(np.float32(s / (4.0 * n)) + np.float32(eps)))
    return np.array(out, np.float32)


# ----------------------------------------------------------------------------- K1
def test_k1_golden_power_map(dev, golden_dir, g1_raw):
    g = np.load(os.path.join(golden_dir, "g1_power.npz"))
    pm = dev.chunk_power(g1_raw)
    assert pm.dtype == np.float32 and pm.shape == (21,)
    np.testing.assert_allclose(pm, g["power_map"], rtol=1e-6)
    np.testing.assert_array_equal(pm, exact_chunk_power(g1_raw, 65536))
    base, thr, ranges = orc.power_threshold(pm)          # host rule of worker.py:241-264
    np.testing.assert_allclose(base, g["baseline"], rtol=1e-6)
    assert [list(r) for r in ranges] == g["ranges"].tolist()


@pytest.mark.parametrize("nbytes", [0, 1, 2, 3, 15, 16, 17, 4095, 65535, 65536, 65537, 200001])
def test_k1_ragged_sizes(dev, nbytes):
    raw = generate(StreamSpec(seed=5), (nbytes + 1) // 2)[:nbytes]
    pm = dev.chunk_power(raw)
    want = orc.chunk_power(raw) if nbytes else np.zeros(0, np.float32)
    assert pm.shape == want.shape
    np.testing.assert_allclose(pm, want, rtol=1e-6, equal_nan=True)
    np.testing.assert_array_equal(pm, exact_chunk_power(raw, 65536))


@pytest.mark.parametrize("chunk_bytes", [2, 30, 1000, 4098, 131072, 4096000])
def test_k1_chunk_sizes(dev, g1_raw, chunk_bytes):
    raw = g1_raw if chunk_bytes < 4096000 else np.tile(g1_raw, 8)[:2 * 4096000 + 777]
    pm = dev.chunk_power(raw, chunk_bytes=chunk_bytes)
    np.testing.assert_array_equal(pm, exact_chunk_power(raw, chunk_bytes))
    np.testing.assert_allclose(pm, orc.chunk_power(raw, chunk_bytes), rtol=1e-6, equal_nan=True)


def test_k1_check_if_jamming_flavour(dev, golden_dir, g1_raw):
    g = np.load(os.path.join(golden_dir, "g1_power.npz"))
    pm = dev.chunk_power(g1_raw, chunk_bytes=orc.CIJ_CHUNK_BYTES, eps=0.0, odd_chunk_zero=True)
    assert pm[-1] == 0.0                                   # odd-sized tail (checkIfJamming.py:12)
    np.testing.assert_allclose(pm.astype(np.float64), g["cij_power"], rtol=1e-6)


def test_k1_threshold_on_device(dev, golden_dir, g1_raw):
    g = np.load(os.path.join(golden_dir, "g1_power.npz"))
    buf = dev.alloc(g1_raw.size).upload(g1_raw)
    n = dev.chunk_count(g1_raw.size, 65536)
    d_pow, d_stats, d_mask = dev.alloc(4 * n), dev.alloc(12), dev.alloc(n)
    dev.chunk_power_dev(buf, g1_raw.size, 65536, d_pow)
    dev.power_threshold_dev(d_pow, n, d_stats, d_mask)
    dev.synchronize()
    stats = d_stats.download(np.float32)
    mask = d_mask.download(np.uint8).astype(bool)
    pm = d_pow.download(np.float32)
    base, thr, _ = orc.power_threshold(pm)
    assert stats[0] == np.float32(base) == np.float32(g["baseline"])
    assert stats[1] == np.float32(thr)
    np.testing.assert_array_equal(mask, pm > thr)
    assert int(stats[2]) == int(mask.sum()) == 4


@pytest.mark.parametrize("n", [1, 2, 3, 20, 21, 257, 1000, 16384, 40000])
def test_percentile_rule_matches_numpy(dev, n):
    rng = np.random.RandomState(n)
    pm = (rng.rand(n).astype(np.float32) * 100 + 50)
    d_pow, d_stats = dev.alloc(4 * n).upload(pm), dev.alloc(12)
    dev.power_threshold_dev(d_pow, n, d_stats)
    dev.synchronize()
    assert d_stats.download(np.float32)[0] == np.percentile(pm, 5)


@pytest.mark.parametrize("n", [7, 1000, 16383, 16385, 33000])
@pytest.mark.parametrize("kind", ["few_values", "constant", "negative_and_zero"])
def test_percentile_rule_ties_and_signs(dev, n, kind):
    """Radix select with heavy ties (run-length flushed histogram), one repeated value, and keys of
    both signs; n straddles the 16 384 register-cached keys (longer maps re-read global memory)."""
    rng = np.random.RandomState(n + len(kind))
    if kind == "few_values":
        pm = rng.randint(0, 5, n).astype(np.float32) * np.float32(0.37) + np.float32(1.5)
    elif kind == "constant":
        pm = np.full(n, 68.72, np.float32)
    else:
        pm = (rng.randn(n) * 3).astype(np.float32)
        pm[::7] = 0.0
    d_pow, d_stats, d_mask = dev.alloc(4 * n).upload(pm), dev.alloc(12), dev.alloc(n)
    dev.power_threshold_dev(d_pow, n, d_stats, d_mask)
    dev.synchronize()
    stats = d_stats.download(np.float32)
    want = np.float32(np.percentile(pm, 5))
    base = want if want > 0 else np.float32(1.0)          # worker.py:243-244
    assert stats[0] == base
    thr = np.float32(base * np.float32(10.0 ** 0.6))
    np.testing.assert_allclose(stats[1], thr, rtol=2e-7)
    np.testing.assert_array_equal(d_mask.download(np.uint8, n).astype(bool), pm > stats[1])
    assert int(stats[2]) == int((pm > stats[1]).sum())


# ----------------------------------------------------------------------------- K2
def rel_err(got, want, floor=1e-12):
    keep = want > floor
    return float(np.max(np.abs(got[keep] - want[keep]) / want[keep]))


@pytest.mark.parametrize("nperseg", [1024, 4096])
def test_k2_golden_welch(dev, golden_dir, g2_raw, nperseg):
    g = np.load(os.path.join(golden_dir, "g2_welch.npz"))
    psd, db = dev.welch(g2_raw, nperseg=nperseg)
    want = g[f"lin_{nperseg}"]
    assert psd.shape == want.shape == (2, nperseg) and psd.dtype == np.float32
    assert rel_err(psd, want) < 1e-4
    np.testing.assert_allclose(db, g[f"db_{nperseg}"], atol=5e-4)


@pytest.mark.parametrize("nperseg", [16, 32, 64, 128, 256, 512, 1024, 2048, 4096])
def test_k2_all_sizes_vs_oracle(dev, nperseg):
    spec = StreamSpec(seed=nperseg, jam_start=30000, jam_end=90000, jam_sigma=30.0,
                      dc_i_q8=600, dc_q_q8=-900)
    chunk = 65536
    raw = generate(spec, 2 * chunk + max(nperseg, 5000))
    psd, db = dev.welch(raw, chunk_samples=chunk, nperseg=nperseg)
    lin, dbo, _ = orc.widmo_waterfall(raw, nperseg=nperseg, chunk_samples=chunk)
    assert psd.shape == lin.shape and psd.shape[0] == 3
    assert rel_err(psd, lin) < 1e-4
    np.testing.assert_allclose(db, dbo, atol=5e-4)
    unshifted, _ = dev.welch(raw, chunk_samples=chunk, nperseg=nperseg, shift=False, want_db=False)
    np.testing.assert_array_equal(np.fft.fftshift(unshifted, axes=1), psd)


def test_k2_unaligned_output_arrays(dev):
    """Output arrays that are only 4-byte aligned take the one-bin-per-thread finalize; same values
    as the 16-byte (float4) path."""
    raw = generate(StreamSpec(seed=31, jam_start=20000, jam_end=70000, jam_sigma=35.0), 150000)
    nperseg, chunk = 1024, 50000
    rows = dev.welch_rows(raw.size, chunk, nperseg)
    d_iq = dev.alloc(raw.size).upload(raw)
    a_psd, a_db = dev.alloc(4 * rows * nperseg), dev.alloc(4 * rows * nperseg)
    u_psd, u_db = dev.alloc(4 * rows * nperseg + 16), dev.alloc(4 * rows * nperseg + 16)
    dev.welch_dev(d_iq, raw.size, chunk, nperseg, 2.048e6, a_psd, a_db)
    dev.welch_dev(d_iq, raw.size, chunk, nperseg, 2.048e6, u_psd.ptr + 4, u_db.ptr + 4)
    dev.synchronize()
    np.testing.assert_array_equal(u_psd.download(np.float32)[1:1 + rows * nperseg], a_psd.download(np.float32))
    np.testing.assert_array_equal(u_db.download(np.float32)[1:1 + rows * nperseg], a_db.download(np.float32))
    lin, _, _ = orc.widmo_waterfall(raw, nperseg=nperseg, chunk_samples=chunk)
    assert rel_err(a_psd.download(np.float32).reshape(rows, nperseg), lin) < 1e-4


def test_k2_partial_chunk_rule(dev):
    nperseg = 1024
    base = generate(StreamSpec(seed=9), 3000)
    # one segment exactly
    psd, _ = dev.welch(base[:2 * nperseg], chunk_samples=2048000, nperseg=nperseg, want_db=False)
    lin, _, _ = orc.widmo_waterfall(base[:2 * nperseg], nperseg=nperseg)
    assert psd.shape == (1, nperseg) and rel_err(psd, lin) < 1e-4
    # one byte short of a segment -> no row (widmo_plot.py:31)
    psd, _ = dev.welch(base[:2 * nperseg - 1], chunk_samples=2048000, nperseg=nperseg, want_db=False)
    assert psd.shape == (0, nperseg)
    with pytest.raises(gpsjam.GpsJamError):
        dev.welch(base, nperseg=1000)


def test_k2_strong_dc_and_tone(dev):
    """Large DC offset (exercises the frequency-domain detrend) plus a strong CW tone."""
    n = 200000
    t = np.arange(n)
    rng = np.random.RandomState(3)
    z = 40 * np.exp(2j * np.pi * 0.1234 * t) + rng.normal(0, 6, n) + 1j * rng.normal(0, 6, n) + (25 - 17j)
    iq = np.empty(2 * n, np.float64)
    iq[0::2], iq[1::2] = z.real, z.imag
    raw = (np.clip(np.trunc(iq), -128, 127) + 128).astype(np.uint8)
    for nperseg in (1024, 4096):
        psd, _ = dev.welch(raw, chunk_samples=n, nperseg=nperseg, want_db=False)
        lin, _, _ = orc.widmo_waterfall(raw, nperseg=nperseg, chunk_samples=n)
        assert rel_err(psd, lin) < 1e-4


# ----------------------------------------------------------------------------- K3
def test_k3_golden_amp_stats(dev, golden_meta, g3_raws):
    for key, (idx, avg) in golden_meta["g3"]["amp_stats"].items():
        k, thr_s = key.split("_")
        st = dev.amp_stats(g3_raws[int(k)], float(thr_s))
        assert st.first_index == idx
        assert st.count == g3_raws[int(k)].size // 2 - idx
        np.testing.assert_allclose(st.mean, avg, rtol=1e-6)
    st = dev.amp_stats(g3_raws[0], 5.0)
    assert st.first_index == -1 and st.count == 0
    st = dev.amp_stats(np.zeros(0, np.uint8), 0.0)
    assert st.first_index == -1


@pytest.mark.parametrize("nsamples", [1, 7, 8, 9, 32767, 32768, 32769, 100001])
def test_k3_sizes(dev, nsamples):
    raw = generate(StreamSpec(seed=31, jam_start=nsamples // 2, jam_end=1 << 40, jam_sigma=50.0), nsamples)
    for thr in (0.0, 0.6):
        st = dev.amp_stats(raw, thr)
        k, avg = orc.rssi_amp_stats(raw, thr)
        if k is None:
            assert st.first_index == -1
        else:
            assert st.first_index == k
            np.testing.assert_allclose(st.mean, avg, rtol=1e-6)


# ----------------------------------------------------------------------------- K4
def test_k4_golden_onset(dev, golden_meta, g4_raws):
    g4 = golden_meta["g4"]
    for raw, want in zip(g4_raws, g4["onset"]):
        assert dev.onset(raw).start_index == want
    assert dev.onset(g4_raws[0][:2 * 200500]).start_index == g4["onset_short"] == -1
    assert dev.onset(g4_raws[0][:2 * 250000]).start_index == g4["onset_none"] == -1
    assert dev.onset(g4_raws[1], 50000, 256, 20.0).start_index == g4["onset_alt"]


@pytest.mark.parametrize("onset_at,window", [(201000, 1000), (208191, 1000), (208192, 1000),
                                             (215000, 37), (230000, 8192)])
def test_k4_tile_boundaries(dev, onset_at, window):
    raw = generate(StreamSpec(seed=44, jam_start=onset_at, jam_end=1 << 40, jam_sigma=70.0), 260000)
    z = orc.tdoa_unpack(raw)
    assert dev.onset(raw, 200000, window, 50.0).start_index == orc.tdoa_onset(z, 200000, window, 50.0)


# ----------------------------------------------------------------------------- K5
def test_k5_golden_lags(dev, golden_meta, g4_raws):
    g4 = golden_meta["g4"]
    onset = g4["onset"]
    pairs = [(0, 1), (0, 2), (1, 2)]
    for n in gi.G4_SLICES:
        own = [r[2 * o:2 * (o + n)] for r, o in zip(g4_raws, onset)]
        lags, peaks = dev.xcorr_lags(own, pairs)
        for (a, b), lag, pk in zip(pairs, lags, peaks):
            assert lag == g4["lags_own_start"][f"{n}_{a}{b}"]
            np.testing.assert_allclose(pk, g4["peaks"][f"own_{n}_{a}{b}"], rtol=1e-4)
        for (a, b) in pairs:
            sl = [g4_raws[a][2 * onset[a]:2 * (onset[a] + n)], g4_raws[b][2 * onset[a]:2 * (onset[a] + n)]]
            lag, pk = dev.xcorr_lags(sl, [(0, 1)])
            assert lag[0] == g4["lags_common_start"][f"{n}_{a}{b}"] == gi.G4_DELAYS[b] - gi.G4_DELAYS[a]
            np.testing.assert_allclose(pk[0], g4["peaks"][f"common_{n}_{a}{b}"], rtol=1e-4)


@pytest.mark.parametrize("n", [1, 2, 100, 4097, 32768, 32769, 70000])
def test_k5_sizes_vs_oracle(dev, n):
    d = 0 if n < 3 else min(n // 3, 37)
    a = generate(StreamSpec(seed=77, antenna=0, delay=0, jam_start=-(1 << 40), jam_end=1 << 40, jam_sigma=45.0), n)
    b = generate(StreamSpec(seed=77, antenna=1, delay=d, jam_start=-(1 << 40), jam_end=1 << 40, jam_sigma=45.0), n)
    lags, peaks = dev.xcorr_lags([a, b], [(0, 1), (1, 0), (0, 0)])
    want01, pk01 = orc.xcorr_lag(orc.tdoa_unpack(b), orc.tdoa_unpack(a))
    want10, _ = orc.xcorr_lag(orc.tdoa_unpack(a), orc.tdoa_unpack(b))
    assert lags[0] == want01 and lags[1] == want10 and lags[2] == 0
    if n >= 100:
        assert lags[0] == d and lags[1] == -d
    np.testing.assert_allclose(peaks[0], pk01, rtol=1e-4)


def test_k5_device_starts_and_invalid(dev, golden_meta, g4_raws):
    """Device-resident captures, start offsets chained from K4 on the device."""
    g4 = golden_meta["g4"]
    n = 50000
    bufs = [dev.alloc(r.size).upload(r) for r in g4_raws]
    d_on = [dev.alloc(32) for _ in bufs]
    d_starts = dev.alloc(8 * 3)
    for b, r, o in zip(bufs, g4_raws, d_on):
        dev.onset_dev(b, r.size, 200000, 1000, 50.0, o)
    dev.synchronize()
    starts = np.array([o.download(np.int64, 1)[0] for o in d_on], np.int64)
    assert starts.tolist() == g4["onset"]
    d_lags, d_peaks = dev.alloc(12), dev.alloc(12)
    d_starts.upload(starts)
    dev.xcorr_lags_dev(bufs, [r.size for r in g4_raws], d_starts, n, [(0, 1), (0, 2), (1, 2)], d_lags, d_peaks)
    dev.synchronize()
    assert d_lags.download(np.int32).tolist() == [g4["lags_own_start"][f"{n}_{k}"] for k in ("01", "02", "12")]
    bad = starts.copy()
    bad[1] = -1                                            # onset not found on antenna 1
    bad[2] = g4_raws[2].size // 2 - n + 1                  # slice would run off the end
    d_starts.upload(bad)
    dev.xcorr_lags_dev(bufs, [r.size for r in g4_raws], d_starts, n, [(0, 1), (0, 2), (0, 0)], d_lags, d_peaks)
    dev.synchronize()
    got = d_lags.download(np.int32).tolist()
    assert got[0] == gpsjam.GJ_LAG_INVALID and got[1] == gpsjam.GJ_LAG_INVALID and got[2] == 0


# ----------------------------------------------------------------------------- synth + histogram
def test_synth_bit_identical(dev):
    spec = StreamSpec(seed=99, antenna=3, delay=-7, jam_start=1000, jam_end=50000, jam_sigma=33.0,
                      dc_i_q8=300, dc_q_q8=-200)
    for n, first in ((100003, 0), (4096, 12345), (5, 0)):
        buf = dev.alloc(2 * n)
        dev.synth_dev(spec, n, buf, first_sample=first)
        dev.synchronize()
        np.testing.assert_array_equal(buf.download(np.uint8), generate(spec, n, first_sample=first))


def test_byte_histogram(dev, g2_raw):
    buf = dev.alloc(g2_raw.size).upload(g2_raw)
    d_hist = dev.alloc(8 * 256)
    dev.byte_histogram_dev(buf, g2_raw.size, 2048000, 1024, 100, d_hist)
    dev.synchronize()
    _, _, hist_samples = orc.widmo_waterfall(g2_raw, nperseg=1024)
    np.testing.assert_array_equal(d_hist.download(np.uint64), np.bincount(hist_samples, minlength=256))


# ----------------------------------------------------------------------------- full size
def test_full_size_1gib_properties(dev):
    """BASELINE config 2 size: 2^30 bytes generated in HBM.  K1 against the integer-exact
    restatement on every chunk; K2 rows against the oracle on three chunks (first, one in the
    jammed span, the ragged last) and against a stand-alone run on those bytes; K3 sum
    against a float64 sum on a 64 MiB prefix."""
    nbytes = 1 << 30
    ns = nbytes // 2
    spec = StreamSpec(seed=1234, antenna=0, jam_start=int(0.4 * ns), jam_end=int(0.7 * ns), jam_sigma=40.0)
    buf = dev.alloc(nbytes)
    dev.synth_dev(spec, ns, buf)
    nchunks = dev.chunk_count(nbytes, 65536)
    rows = dev.welch_rows(nbytes, 2048000, 4096)
    assert nchunks == 16384 and rows == 263
    d_pow, d_psd, d_amp = dev.alloc(4 * nchunks), dev.alloc(4 * rows * 4096), dev.alloc(32)
    dev.chunk_power_dev(buf, nbytes, 65536, d_pow)
    dev.welch_dev(buf, nbytes, 2048000, 4096, 2.048e6, d_psd)
    dev.amp_stats_dev(buf, nbytes, 0.0, d_amp)
    dev.synchronize()
    raw = buf.download(np.uint8)
    np.testing.assert_array_equal(raw[:200000], generate(spec, 100000))
    pm = d_pow.download(np.float32)
    np.testing.assert_array_equal(pm, exact_chunk_power(raw, 65536))
    base, thr, ranges = orc.power_threshold(pm)
    assert len(ranges) == 1
    assert abs(ranges[0][0] - 2 * int(0.4 * ns)) <= 65536 and abs(ranges[0][1] - 2 * int(0.7 * ns)) <= 65536
    psd = d_psd.download(np.float32).reshape(rows, 4096)
    assert np.all(np.isfinite(psd)) and np.all(psd > 0)
    for c in (0, 131, 262):
        piece = raw[c * 4096000:(c + 1) * 4096000]
        lin, _ = orc.widmo_chunk_psd_db(piece, nperseg=4096)
        assert rel_err(psd[c], lin) < 1e-4
        alone, _ = dev.welch(piece, nperseg=4096, want_db=False)
        np.testing.assert_allclose(alone[0], psd[c], rtol=1e-5)   # other split, other summation order
    amp = np.frombuffer(d_amp.download(np.uint8).tobytes(), dtype=[("i", "<i8"), ("c", "<u8"), ("s", "<f8"),
                                                                   ("m", "<f4"), ("r", "<f4")])[0]
    assert amp["i"] == 0 and amp["c"] == ns
    v = 2.0 * raw[:1 << 26].astype(np.float64) - 255.0
    ref_part = np.sqrt(v[0::2] ** 2 + v[1::2] ** 2).sum() / 255.0
    st = dev.amp_stats(raw[:1 << 26], 0.0)
    np.testing.assert_allclose(st.sum, ref_part, rtol=1e-7)


# ----------------------------------------------------------------------------- fused stream scan
@pytest.mark.parametrize("nbytes,chunk,thr", [(20 * 65536 + 24691, 65536, 0.0), (20 * 65536 + 24691, 131072, 0.45),
                                              (65536 * 7, 65536, 0.1), (65536 * 3 + 254, 65536, 0.0),
                                              (65536 * 3 + 258, 65536, 0.2), (600001, 65536, 0.0),
                                              (400000, 1000, 0.0), (2, 65536, 0.0), (3, 65536, 0.0)])
def test_stream_scan_equals_separate_kernels(dev, nbytes, chunk, thr):
    """gj_stream_scan_dev (one pass) against K1, K3, K4 called one after the other."""
    n = (nbytes + 1) // 2
    raw = generate(StreamSpec(seed=nbytes & 0xffff, jam_start=220000, jam_end=1 << 40, jam_sigma=60.0), n)[:nbytes]
    buf = dev.alloc(max(nbytes, 16)).upload(raw)
    nch = dev.chunk_count(nbytes, chunk)
    out = {}
    for mode in ("fused", "separate"):
        d_pow, d_amp, d_on = dev.alloc(4 * max(nch, 1)), dev.alloc(32), dev.alloc(32)
        if mode == "fused":
            dev.stream_scan_dev(buf, nbytes, chunk, d_pow, thr, d_amp, 200000, 1000, 50.0, d_on)
        else:
            dev.chunk_power_dev(buf, nbytes, chunk, d_pow)
            dev.amp_stats_dev(buf, nbytes, thr, d_amp)
            dev.onset_dev(buf, nbytes, 200000, 1000, 50.0, d_on)
        dev.synchronize()
        amp = np.frombuffer(d_amp.download(np.uint8).tobytes(), dtype=[("i", "<i8"), ("c", "<u8"), ("s", "<f8"),
                                                                       ("m", "<f4"), ("r", "<f4")])[0]
        out[mode] = (d_pow.download(np.float32, nch), amp, int(d_on.download(np.int64, 1)[0]))
    np.testing.assert_array_equal(out["fused"][0], out["separate"][0])
    np.testing.assert_array_equal(out["fused"][0], exact_chunk_power(raw, chunk))
    assert out["fused"][1]["i"] == out["separate"][1]["i"] and out["fused"][1]["c"] == out["separate"][1]["c"]
    np.testing.assert_allclose(out["fused"][1]["s"], out["separate"][1]["s"], rtol=1e-7)   # f32 partial sums group differently
    assert out["fused"][2] == out["separate"][2]
    even = raw[:2 * (nbytes // 2)]          # the reference itself cannot unpack an odd-length file
    k, avg = orc.rssi_amp_stats(even, thr)
    if k is not None:
        assert out["fused"][1]["i"] == k
        np.testing.assert_allclose(out["fused"][1]["m"], avg, rtol=1e-6)
    z = orc.tdoa_unpack(even)
    assert out["fused"][2] == orc.tdoa_onset(z)
    # the integer-exact restatement of the header's contract (tests/exact_restatement.py), independent of either entry point
    assert out["fused"][2] == ex.onset(raw)["start"]
    want = ex.amp_stats(raw, thr)
    assert (out["fused"][1]["i"], out["fused"][1]["c"]) == (want["first"], want["count"])
    np.testing.assert_allclose(out["fused"][1]["s"], want["sum"], rtol=2e-7)


@pytest.mark.parametrize("tail_bytes,back", [(40000 + 346, 2500), (13 * 1024 + 712, 6000), (65536 - 1024 + 40, 3100),
                                             (50 * 1024 + 1000, 7000), (20 * 1024 + 16, 2048)])
def test_stream_scan_burst_in_partial_last_tile(dev, tail_bytes, back):
    """Capture whose length is not a multiple of 1 KiB, last 64-KiB tile partly filled (> 12 KiB),
    burst starting a few thousand samples before the end: the 512-sample screening sums of the last
    tile must be complete (a wave that splits at the end of the stream used to leave three of them
    under-counted), so the fused pass must find the same onset as K4 alone and the oracle."""
    nbytes = 9 * 65536 + tail_bytes
    ns = nbytes // 2
    start = ns - back
    raw = generate(StreamSpec(seed=77 + back, jam_start=start, jam_end=1 << 40, jam_sigma=90.0), ns)[:nbytes]
    buf = dev.alloc(nbytes).upload(raw)
    nch = dev.chunk_count(nbytes, 65536)
    d_pow, d_amp, d_on, d_on2 = dev.alloc(4 * nch), dev.alloc(32), dev.alloc(32), dev.alloc(32)
    dev.stream_scan_dev(buf, nbytes, 65536, d_pow, 0.0, d_amp, 200000, 1000, 50.0, d_on)
    dev.onset_dev(buf, nbytes, 200000, 1000, 50.0, d_on2)
    dev.synchronize()
    want = orc.tdoa_onset(orc.tdoa_unpack(raw[:2 * ns]))
    assert want > 0, "the test input must contain a detectable onset"
    assert int(d_on2.download(np.int64, 1)[0]) == want
    assert int(d_on.download(np.int64, 1)[0]) == want
    np.testing.assert_array_equal(d_pow.download(np.float32, nch), exact_chunk_power(raw, 65536))
