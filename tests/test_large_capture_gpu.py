"""A capture larger than 4 GiB (an hour of 2.048 MS/s I/Q is 14.7 GB; a GPU holds 288 GB): every byte offset past 2^32
and every sample index past 2^31 must be computed in 64 bits.  5 GiB + a ragged tail, generated in HBM, burst starting
at sample 2.3e9 (byte 4.6e9).  No oracle can chew 5 GiB in seconds, so the whole-capture results are checked against
the same kernels run on three pieces shorter than 2^31 bytes (cut where the 64-KiB power chunks and the 1-s Welch
chunks both end) and against the oracle on the windows that decide the results."""
import numpy as np
import pytest

from gpsjam.synth import StreamSpec, generate
from oracle import gpsjam_oracle as orc

pytestmark = pytest.mark.gpu

NBYTES = 5 * (1 << 30) + 123456
NS = NBYTES // 2
JAM = 2_300_000_000                      # sample index > 2^31
CUT = 8_192_000                          # lcm(65 536, 4 096 000) bytes
P1, P2 = 262 * CUT, 524 * CUT            # 2.146e9 and 4.293e9 bytes: the third piece straddles byte 2^32
AMP_DT = [("i", "<i8"), ("c", "<u8"), ("s", "<f8"), ("m", "<f4"), ("r", "<f4")]


def _amp(buf):
    return np.frombuffer(buf.download(np.uint8, 32).tobytes(), dtype=AMP_DT)[0]


def test_capture_beyond_4gib(dev):
    info = dev.info()
    if info["hbm_bytes"] < 24 * (1 << 30):
        pytest.skip("needs a GPU with room for a 5 GiB capture")
    spec = StreamSpec(seed=4321, antenna=0, jam_start=JAM, jam_end=1 << 40, jam_sigma=70.0)
    buf = dev.alloc(NBYTES + 256)
    dev.synth_dev(spec, NS, buf)
    dev.synchronize()
    # the generator itself: bytes far beyond 2^32 equal the numpy generator's
    for first in (0, (1 << 31) - 50, JAM - 100, NS - 4000):
        got = buf.download(np.uint8, 2 * 4000, offset=2 * first)
        np.testing.assert_array_equal(got, generate(spec, 4000, first_sample=first))

    nch, rows = dev.chunk_count(NBYTES, 65536), dev.welch_rows(NBYTES, 2048000, 4096)
    assert nch == 81922 and rows == 1311
    d_pow, d_psd, d_amp, d_on = dev.alloc(4 * nch), dev.alloc(4 * rows * 4096), dev.alloc(32), dev.alloc(32)
    dev.chunk_power_dev(buf, NBYTES, 65536, d_pow)
    dev.welch_dev(buf, NBYTES, 2048000, 4096, 2.048e6, d_psd)
    dev.amp_stats_dev(buf, NBYTES, 0.9, d_amp)                 # |z| > 0.9 happens only inside the burst
    dev.onset_dev(buf, NBYTES, 200000, 1000, 50.0, d_on)
    dev.synchronize()
    pm = d_pow.download(np.float32, nch)
    psd = d_psd.download(np.float32, rows * 4096).reshape(rows, 4096)
    amp, onset = _amp(d_amp), int(d_on.download(np.int64, 1)[0])

    # the fused pass of the pipeline over the same 5 GiB
    f_pow, f_amp, f_on = dev.alloc(4 * nch), dev.alloc(32), dev.alloc(32)
    dev.stream_scan_dev(buf, NBYTES, 65536, f_pow, 0.9, f_amp, 200000, 1000, 50.0, f_on)
    dev.synchronize()
    np.testing.assert_array_equal(f_pow.download(np.float32, nch), pm)
    fa = _amp(f_amp)
    assert fa["i"] == amp["i"] and fa["c"] == amp["c"] and int(f_on.download(np.int64, 1)[0]) == onset
    np.testing.assert_allclose(fa["s"], amp["s"], rtol=1e-7)

    # pieces shorter than 2^31 bytes give the same rows / chunks
    pieces = [(0, P1), (P1, P2), (P2, NBYTES)]
    pm_parts, psd_parts, sums, counts, firsts = [], [], [], [], []
    for lo, hi in pieces:
        n = hi - lo
        c, r = dev.chunk_count(n, 65536), dev.welch_rows(n, 2048000, 4096)
        p_pow, p_psd, p_amp = dev.alloc(4 * c), dev.alloc(4 * r * 4096), dev.alloc(32)
        dev.chunk_power_dev(buf.ptr + lo, n, 65536, p_pow)
        dev.welch_dev(buf.ptr + lo, n, 2048000, 4096, 2.048e6, p_psd)
        dev.amp_stats_dev(buf.ptr + lo, n, 0.9, p_amp)
        dev.synchronize()
        pm_parts.append(p_pow.download(np.float32, c))
        psd_parts.append(p_psd.download(np.float32, r * 4096).reshape(r, 4096))
        a = _amp(p_amp)
        firsts.append(int(a["i"]) + lo // 2 if a["i"] >= 0 else -1)
        counts.append(int(a["c"]))
        sums.append(float(a["s"]))
        for b in (p_pow, p_psd, p_amp):
            b.free()
    np.testing.assert_array_equal(np.concatenate(pm_parts), pm)
    whole = np.concatenate(psd_parts)
    assert whole.shape == psd.shape
    np.testing.assert_allclose(psd, whole, rtol=1e-5)          # other split of a chunk over workgroups, other summation order
    # K3: first crossing inside the burst (third piece), everything after it counted
    assert firsts[0] == -1 and firsts[1] == -1 and firsts[2] == amp["i"] and amp["i"] >= JAM
    assert amp["c"] == NS - amp["i"] == counts[2]
    np.testing.assert_allclose(amp["s"], sums[2], rtol=1e-9)

    # K1 / K2 against the oracle where it matters: the chunks around byte 2^32, the burst edge and the ragged end
    for chunk_idx in (0, (1 << 32) // 65536 - 1, (1 << 32) // 65536, 2 * JAM // 65536, nch - 1):
        lo = chunk_idx * 65536
        piece = buf.download(np.uint8, min(65536, NBYTES - lo), offset=lo)
        np.testing.assert_allclose(pm[chunk_idx], orc.chunk_power(piece)[0], rtol=1e-6)
    for row in (0, (1 << 32) // 4096000, 2 * JAM // 4096000, rows - 1):
        lo = row * 4096000
        piece = buf.download(np.uint8, min(4096000, NBYTES - lo), offset=lo)
        lin, _ = orc.widmo_chunk_psd_db(piece, nperseg=4096)
        keep = lin > 1e-12
        assert float(np.max(np.abs(psd[row][keep] - lin[keep]) / lin[keep])) < 1e-4

    # K4: the oracle on [noise head | window around the burst edge]; indices map back to the capture
    head = buf.download(np.uint8, 2 * 200000)
    w0 = JAM - 6000
    win = buf.download(np.uint8, 2 * 12000, offset=2 * w0)
    red = orc.tdoa_onset(orc.tdoa_unpack(np.concatenate([head, win])), 200000, 1000, 50.0)
    assert red > 200000 and onset == red - 200000 + w0 and abs(onset - JAM) < 1000 and onset > (1 << 31)

    # the TDOA slot and K5 at a start beyond 2^31 samples: a second slice = the same capture cut 23 samples earlier
    n = 1 << 17
    sb = dev.tdoa_slot_bytes(n)
    d_slot = dev.alloc(2 * sb)
    dev.tdoa_slot_dev(buf, NBYTES, d_on, n, d_slot)
    d_start2 = dev.alloc(8).upload(np.array([onset - 23], np.int64))
    dev.tdoa_slot_dev(buf, NBYTES, d_start2, n, d_slot.ptr + sb)
    d_res = dev.alloc(64)
    dev.xcorr_slots_dev(d_slot, sb, 2, n, [(0, 1)], d_res, d_res.ptr + 16, d_res.ptr + 32)
    dev.synchronize()
    slot = d_slot.download(np.uint8, sb)
    assert slot[:16].view(np.int64).tolist() == [0, onset]
    np.testing.assert_array_equal(slot[16:16 + 2 * n], buf.download(np.uint8, 2 * n, offset=2 * onset))
    assert int(d_res.download(np.int32, 1)[0]) == 23           # slot 1 starts 23 samples earlier: it holds slot 0 delayed by 23
    # the same through the start-array form on the resident capture
    d_starts = dev.alloc(16).upload(np.array([onset, onset - 23], np.int64))
    dev.xcorr_lags_dev([buf, buf], [NBYTES, NBYTES], d_starts, n, [(0, 1)], d_res, d_res.ptr + 16, d_res.ptr + 32)
    dev.synchronize()
    assert int(d_res.download(np.int32, 1)[0]) == 23
    for b in (buf, d_pow, d_psd, d_amp, d_on, f_pow, f_amp, f_on, d_slot, d_start2, d_res, d_starts):
        b.free()
