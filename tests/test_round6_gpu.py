"""Round-6 additions on the GPU: K3 / K4 ALONE (gj_amp_stats_dev, gj_onset_dev) are the fused pass + tail since the round-2
chains behind them were retired -- checked against the integer-exact restatement of the header's contract and the oracle,
on aligned and unaligned captures; the power map's rule-decided last entry is written AHEAD of the tail launch (ADVICE r05:
the threshold workgroup used to read a word another workgroup of the same launch wrote)."""
import numpy as np
import pytest

import gpsjam
from gpsjam import _ffi
from gpsjam.synth import StreamSpec, generate
from oracle import gpsjam_oracle as orc
import exact_restatement as ex

pytestmark = pytest.mark.gpu

NOISE, WINDOW, FACTOR = 200000, 1000, 50.0
GJ_ERR_INVALID, GJ_ERR_UNSUPPORTED = -1, -5
ONSET_T = np.dtype([("start", "<i8"), ("noise", "<f4"), ("thr", "<f4"), ("hit", "<f4"), ("before", "<f4"), ("guard", "<i8")])
AMP_T = np.dtype([("i", "<i8"), ("c", "<u8"), ("s", "<f8"), ("m", "<f4"), ("r", "<f4")])


def _alone(dev, ptr, nbytes, thr, noise=NOISE, window=WINDOW):
    d_amp, d_on = dev.alloc(32), dev.alloc(32)
    dev.amp_stats_dev(ptr, nbytes, thr, d_amp)
    dev.onset_dev(ptr, nbytes, noise, window, FACTOR, d_on)
    dev.synchronize()
    out = d_amp.download(np.uint8, 32).tobytes(), d_on.download(np.uint8, 32).tobytes()
    d_amp.free(), d_on.free()
    return out


def _check_exact(amp_bytes, on_bytes, raw, thr, noise=NOISE, window=WINDOW):
    a, o = np.frombuffer(amp_bytes, AMP_T)[0], np.frombuffer(on_bytes, ONSET_T)[0]
    want_a, want_o = ex.amp_stats(raw, thr), ex.onset(raw, noise, window, FACTOR)
    assert (a["i"], a["c"]) == (want_a["first"], want_a["count"])
    np.testing.assert_allclose(a["s"], want_a["sum"], rtol=2e-7)
    np.testing.assert_allclose(a["m"], want_a["mean"], rtol=2e-7)
    for f in ("start", "guard", "noise", "thr", "hit"):
        assert o[f] == want_o[f], (f, o, want_o)


@pytest.mark.parametrize("nbytes,thr,jam", [(20 * 65536 + 24691, 0.0, 220000), (20 * 65536 + 24690, 0.45, 220000), (7 * 65536, 0.1, 210000),
                                            (2 * (NOISE + WINDOW), 0.0, 1 << 40), (2 * (NOISE + WINDOW) - 2, 0.3, 1 << 40),
                                            (3, 0.0, 0), (2, 0.0, 0), (1, 0.0, 0), (0, 0.0, 0), (40_960_000, 0.0, 9_000_000)])
def test_k3_k4_alone_are_the_fused_pass(dev, nbytes, thr, jam):
    """gj_amp_stats_dev / gj_onset_dev against the exact restatement and the oracle, and byte-equal to the records
    gj_stream_scan_dev writes for the same capture (same tiles, same integer sums, same summation order)."""
    n = (nbytes + 1) // 2
    raw = generate(StreamSpec(seed=(nbytes & 0xffff) + 1, jam_start=jam, jam_end=1 << 40, jam_sigma=60.0), max(n, 1))[:nbytes]
    buf = dev.alloc(max(nbytes, 16)).upload(raw) if nbytes else dev.alloc(16)
    amp_b, on_b = _alone(dev, buf, nbytes, thr)
    _check_exact(amp_b, on_b, raw, thr)
    if 0 < nbytes <= 3_000_000:
        even = raw[:2 * (nbytes // 2)]
        assert np.frombuffer(on_b, ONSET_T)[0]["start"] == orc.tdoa_onset(orc.tdoa_unpack(even))
        k, avg = orc.rssi_amp_stats(even, thr)
        a = np.frombuffer(amp_b, AMP_T)[0]
        assert a["i"] == (-1 if k is None else k)
        if k is not None:
            np.testing.assert_allclose(a["m"], avg, rtol=1e-6)
    nch = dev.chunk_count(nbytes, 65536)
    d_pow, d_amp, d_on = dev.alloc(4 * max(nch, 1)), dev.alloc(32), dev.alloc(32)
    dev.stream_scan_dev(buf, nbytes, 65536, d_pow, thr, d_amp, NOISE, WINDOW, FACTOR, d_on)
    dev.synchronize()
    assert d_amp.download(np.uint8, 32).tobytes() == amp_b
    fused_on = np.frombuffer(d_on.download(np.uint8, 32).tobytes(), ONSET_T)[0]
    alone_on = np.frombuffer(on_b, ONSET_T)[0]
    for f in ("start", "guard", "noise", "thr", "hit", "before"):
        assert fused_on[f] == alone_on[f], f
    np.testing.assert_array_equal(d_pow.download(np.float32, nch), ex.chunk_power(raw, 65536))
    for b in (buf, d_pow, d_amp, d_on):
        b.free()


@pytest.mark.parametrize("off", [1, 2, 6, 14, 15, 16, 4098])
def test_k3_k4_alone_on_captures_that_are_not_16_byte_aligned(dev, off):
    """A caller's pointer into the middle of a buffer: any alignment is accepted (the capture is copied to an aligned place
    in the workspace first) and the results are those of the same bytes at an aligned address -- exact restatement, oracle
    index, and byte-equal records."""
    nbytes = 9 * 65536 + 12346
    raw = generate(StreamSpec(seed=600 + off, jam_start=230_000, jam_end=1 << 40, jam_sigma=70.0), (nbytes + off) // 2 + 1)
    raw = raw[:nbytes + off]
    buf = dev.alloc(nbytes + off + 16).upload(raw)
    piece = raw[off:off + nbytes]
    amp_b, on_b = _alone(dev, buf.ptr + off, nbytes, 0.2)
    _check_exact(amp_b, on_b, piece, 0.2)
    assert np.frombuffer(on_b, ONSET_T)[0]["start"] == orc.tdoa_onset(orc.tdoa_unpack(piece[:2 * (nbytes // 2)]))
    aligned = dev.alloc(nbytes + 16).upload(piece)
    amp_a, on_a = _alone(dev, aligned, nbytes, 0.2)
    assert (amp_a, on_a) == (amp_b, on_b)
    # the chunk sizes the fused pass does not take run K1 alone + the same pass: unaligned there too
    nch = dev.chunk_count(nbytes, 1000)
    d_pow, d_amp, d_on = dev.alloc(4 * nch), dev.alloc(32), dev.alloc(32)
    dev.stream_scan_dev(buf.ptr + off, nbytes, 1000, d_pow, 0.2, d_amp, NOISE, WINDOW, FACTOR, d_on)
    dev.synchronize()
    assert d_amp.download(np.uint8, 32).tobytes() == amp_b
    assert np.frombuffer(d_on.download(np.uint8, 32).tobytes(), ONSET_T)[0]["start"] == np.frombuffer(on_b, ONSET_T)[0]["start"]
    np.testing.assert_array_equal(d_pow.download(np.float32, nch), ex.chunk_power(piece, 1000))
    for b in (buf, aligned, d_pow, d_amp, d_on):
        b.free()


def test_k4_alone_keeps_its_argument_checks(dev):
    buf = dev.alloc(1 << 20)
    d_on = dev.alloc(32)
    for noise, window in ((0, 1000), (200000, 0), (-1, 5)):
        with pytest.raises(gpsjam.GpsJamError) as e:
            dev.onset_dev(buf, 1 << 20, noise, window, FACTOR, d_on)
        assert e.value.status == GJ_ERR_INVALID
    with pytest.raises(gpsjam.GpsJamError) as e:
        dev.onset_dev(buf, 1 << 20, 1000, 8193, FACTOR, d_on)
    assert e.value.status == GJ_ERR_UNSUPPORTED
    buf.free(), d_on.free()


@pytest.mark.parametrize("nbytes,chunk,flags", [(21 * 65536 + 1, 65536, 0), (21 * 65536 + 1, 65536, _ffi.GJ_CP_ODD_CHUNK_ZERO),
                                                (10 * 131072 + 1, 131072, 0), (10 * 131072 + 1, 131072, _ffi.GJ_CP_ODD_CHUNK_ZERO),
                                                (10 * 131072 + 65537, 131072, _ffi.GJ_CP_ODD_CHUNK_ZERO), (1, 65536, 0),
                                                (21 * 65536 + 4097, 65536, _ffi.GJ_CP_ODD_CHUNK_ZERO)])
def test_tail_threshold_with_a_rule_decided_last_chunk(dev, nbytes, chunk, flags):
    """a capture of k chunks + 1 byte (or with the odd-chunk rule on an odd tail) has a last power-map
    entry that no tile of the pass writes; the tail's amplitude workgroup used to write it while the threshold workgroup
    of the SAME launch read the map.  The word is now written ahead of the tail.  The power map is pre-filled with
    sentinels that would move the percentile (very low) or the count (very high) if they were read; baseline / threshold /
    count / mask must equal gj_power_threshold_dev on the finished map and numpy's rule on the exact map."""
    raw = generate(StreamSpec(seed=nbytes & 0xffff, jam_start=400_000, jam_end=1 << 40, jam_sigma=60.0), (nbytes + 1) // 2)[:nbytes]
    buf = dev.alloc(nbytes + 16).upload(raw)
    nch = dev.chunk_count(nbytes, chunk)
    want_pm = ex.chunk_power(raw, chunk, odd_chunk_zero=bool(flags))
    assert want_pm.size == nch
    d_st2, d_mask2 = dev.alloc(16), dev.alloc(nch)
    for sentinel in (-1e30, 1e30):
        d_pow, d_st, d_mask, d_amp, d_on = dev.alloc(4 * nch), dev.alloc(16), dev.alloc(nch), dev.alloc(32), dev.alloc(32)
        for _ in range(3):
            d_pow.upload(np.full(nch, sentinel, np.float32))
            dev.capture_scan_dev(buf, nbytes, chunk, d_pow, 0.0, d_amp, NOISE, WINDOW, FACTOR, d_on, d_stats=d_st, d_mask=d_mask,
                                 flags=flags)
            dev.synchronize()
            pm = d_pow.download(np.float32, nch)
            np.testing.assert_array_equal(pm, want_pm)
            dev.power_threshold_dev(d_pow, nch, d_st2, d_mask2)
            dev.synchronize()
            st, st2 = d_st.download(np.float32, 3), d_st2.download(np.float32, 3)
            np.testing.assert_array_equal(st, st2)
            np.testing.assert_array_equal(d_mask.download(np.uint8, nch), d_mask2.download(np.uint8, nch))
            if not np.isnan(want_pm).any():
                base, thr, _ = orc.power_threshold(want_pm)
                assert st[0] == np.float32(base) and st[2] == np.count_nonzero(want_pm > st[1])
        for b in (d_pow, d_st, d_mask, d_amp, d_on):
            b.free()
    for b in (buf, d_st2, d_mask2):
        b.free()



# ----------------------------------------------------------------------------- seeded sweep of K3 / K4 alone
@pytest.mark.parametrize("case", list(range(16)))
def test_sweep_k3_k4_alone(dev, case):
    """Sizes, alignments, thresholds, noise spans, windows and burst positions nobody hand-picked: gj_amp_stats_dev and
    gj_onset_dev on a pointer at a random offset into a buffer, against the exact restatement (bit for bit) and the oracle
    (index exact, mean 1e-6)."""
    r = np.random.RandomState(6000 + case)
    off = int(r.choice([0, 0, 16, 2, 6, 1, 7, 30, 65536, 65538]))
    nbytes = int(r.randint(3, 40)) * 65536 + int(r.choice([0, 1, 2, 254, 1023, 24691, 65535]))
    noise = int(r.choice([1, 511, 512, 20_000, 100_000, 200_000]))
    window = int(r.choice([1, 8, 256, 1000, 1001, 4096]))
    thr = float(r.choice([0.0, 0.05, 0.2, 0.6, 1.5]))
    ns = nbytes // 2
    jam = int(ns * r.uniform(0.3, 0.9)) if r.rand() < 0.8 else 1 << 40
    raw = generate(StreamSpec(seed=7000 + case, jam_start=jam, jam_end=1 << 40, jam_sigma=float(r.uniform(40, 90))), (nbytes + off) // 2 + 1)
    raw = raw[:nbytes + off]
    buf = dev.alloc(nbytes + off + 16).upload(raw)
    piece = raw[off:off + nbytes]
    amp_b, on_b = _alone(dev, buf.ptr + off, nbytes, thr, noise=noise, window=window)
    _check_exact(amp_b, on_b, piece, thr, noise=noise, window=window)
    even = piece[:2 * ns]
    got_on = np.frombuffer(on_b, ONSET_T)[0]
    if window <= 1001 or ns <= 600_000:                       # the oracle's np.convolve is O(n * window)
        want = orc.tdoa_onset(orc.tdoa_unpack(even), noise, window, FACTOR)
        # exact integers here, float32 sums there: identical unless the decision fell inside the rounding band, which the
        # record says (guard_index != start_index or a margin below 1e-6)
        if got_on["guard"] == got_on["start"] and (got_on["start"] < 0 or got_on["hit"] >= 1e-6):
            assert got_on["start"] == want, (got_on, want)
    k, avg = orc.rssi_amp_stats(even, thr)
    a = np.frombuffer(amp_b, AMP_T)[0]
    assert a["i"] == (-1 if k is None else k)
    if k is not None:
        np.testing.assert_allclose(a["m"], avg, rtol=1e-6)
    buf.free()
