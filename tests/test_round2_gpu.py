"""Round-2 additions of the C-ABI on a live GPU: resident captures (one upload per file), TDOA
slots, decision margins of K4 / K5 with constructed near-ties, and the RCCL communicator with one
rank (N > 1 needs N GPUs: covered on CPU by the gloo test and on one GPU by the two-rank
rehearsal; the RCCL transport itself first runs in the driver's multi-GPU bench)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import gpsjam
from gpsjam import _ffi, sharded
from gpsjam.synth import StreamSpec, generate
from oracle import gpsjam_oracle as orc

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(os.path.dirname(HERE), "gps-jamming_amd")
for p in (os.path.join(PKG, "skrypty"), os.path.join(PKG, "GpsJammerApp", "app")):
    if p not in sys.path:
        sys.path.insert(0, p)


# ----------------------------------------------------------------------------- resident captures
def test_capture_from_file_and_array_equals_host_path(dev, tmp_path):
    n = 1_500_000
    raw = generate(StreamSpec(seed=41, jam_start=700000, jam_end=1 << 40, jam_sigma=50.0), n)
    path = tmp_path / "cap.bin"
    raw.tofile(path)
    before = gpsjam.Capture.uploads
    with dev.capture(str(path)) as cap, dev.capture(raw) as cap2:
        assert gpsjam.Capture.uploads == before + 2
        assert cap.nbytes == cap2.nbytes == raw.size and cap.nsamples == n
        np.testing.assert_array_equal(cap.download(), raw)
        np.testing.assert_array_equal(cap2.download(1000, 4096), raw[1000:5096])
        for c in (cap, cap2):
            np.testing.assert_array_equal(dev.chunk_power(c), dev.chunk_power(raw))
            np.testing.assert_array_equal(dev.chunk_power(c, 131072, 0.0, True), dev.chunk_power(raw, 131072, 0.0, True))
            a, adb = dev.welch(c, chunk_samples=500000, nperseg=1024)
            b, bdb = dev.welch(raw, chunk_samples=500000, nperseg=1024)
            np.testing.assert_array_equal(a, b)
            np.testing.assert_array_equal(adb, bdb)
            a, none = dev.welch(c, chunk_samples=500000, nperseg=4096, want_db=False, shift=False)
            assert none is None
            np.testing.assert_array_equal(a, dev.welch(raw, chunk_samples=500000, nperseg=4096, shift=False)[0])
            sa, sb = dev.amp_stats(c, 0.2), dev.amp_stats(raw, 0.2)
            assert (sa.first_index, sa.count, sa.sum, sa.mean) == (sb.first_index, sb.count, sb.sum, sb.mean)
            oa, ob = dev.onset(c), dev.onset(raw)
            assert (oa.start_index, oa.noise_power, oa.threshold) == (ob.start_index, ob.noise_power, ob.threshold)
            assert oa.start_index == orc.tdoa_onset(orc.tdoa_unpack(raw))
        # everything above ran on two uploads
        assert gpsjam.Capture.uploads == before + 2
        hist = dev.byte_histogram(cap, 500000, 1024, 100)
        _, _, samples = orc.widmo_waterfall(raw, nperseg=1024, chunk_samples=500000)
        np.testing.assert_array_equal(hist, np.bincount(samples, minlength=256))
        # lags between resident captures: no slice leaves HBM
        on = dev.onset(cap).start_index
        lags, peaks, margins = dev.xcorr_lags_at([cap, cap2, cap], [on, on + 3, -1], 1 << 15, [(0, 1), (0, 2)],
                                                 want_margins=True)
        assert lags.tolist() == [-3, gpsjam.GJ_LAG_INVALID] and margins[0] > 0.5
    # offsets / partial reads of a file; empty file
    with dev.capture(str(path), offset=4096, max_bytes=200000) as part:
        np.testing.assert_array_equal(part.download(), raw[4096:204096])
    empty = tmp_path / "empty.bin"
    empty.write_bytes(b"")
    with dev.capture(str(empty)) as e:
        assert e.nbytes == 0 and dev.chunk_power(e).size == 0 and dev.amp_stats(e, 0.0).first_index == -1
    with pytest.raises(gpsjam.GpsJamError):
        dev.capture(str(tmp_path / "missing.bin"))


def test_large_capture_goes_through_the_bounce_buffers(dev, tmp_path):
    """Large uploads (every staged copy from 4 MiB up) use the pinned bounce pipeline: pieces sized to the capture, eight fill threads."""
    n = 40_000_000                                          # 80 MB
    raw = np.arange(2 * n, dtype=np.uint32).astype(np.uint8)
    raw[1::7] ^= 0x5a
    path = tmp_path / "big.bin"
    raw.tofile(path)
    with dev.capture(str(path)) as a, dev.capture(raw) as b:
        for c in (a, b):
            for off in (0, (32 << 20) - 8, (64 << 20) - 100, raw.size - 5000):
                np.testing.assert_array_equal(c.download(off, 5000), raw[off:off + 5000])
        np.testing.assert_array_equal(dev.chunk_power(a), dev.chunk_power(b))


def test_resident_capture_cache_one_upload_per_file(tmp_path, monkeypatch):
    """The worker's scan -> RSSI triangulation flow and the PSD script on the same files: one
    host->device pass per file."""
    monkeypatch.setattr(gpsjam, "_default", None)
    gpsjam.release_resident()
    import triangulateRSSI
    import widmo_plot
    raws = [generate(StreamSpec(seed=50 + a, antenna=a, jam_start=0, jam_end=1 << 40, jam_sigma=s), 400000)
            for a, s in enumerate((60.0, 40.0, 25.0))]
    paths = []
    for k, r in enumerate(raws):
        p = tmp_path / f"a{k}.bin"
        r.tofile(p)
        paths.append(str(p))
    before = gpsjam.Capture.uploads
    cap = gpsjam.resident_capture(paths[0])
    pm = cap.dev.chunk_power(cap)                                           # the worker's power scan
    res = triangulateRSSI.triangulate_jammer_location(paths, threshold=0.0)  # reuses file 0, uploads 1 and 2
    out = widmo_plot.analyze_full_file(paths[0], chunk_size=100000)          # reuses file 0 again
    res2 = triangulateRSSI.triangulate_jammer_location(paths, threshold=0.1)
    assert gpsjam.Capture.uploads == before + 3
    assert res["success"] and res2["success"] and out["spectrogram"].shape == (4, 1024)
    np.testing.assert_allclose(pm, orc.chunk_power(raws[0]), rtol=1e-6)
    want = orc.triangulate(raws, threshold=0.0)
    np.testing.assert_allclose(res["distances"], want["distances"], rtol=1e-5)
    # a rewritten file is a different capture
    raws[0][:1000] = 200
    raws[0].tofile(paths[0])
    os.utime(paths[0], ns=(1, 1))
    cap_b = gpsjam.resident_capture(paths[0])
    assert gpsjam.Capture.uploads == before + 4 and cap_b.download(0, 10).tolist() == [200] * 10
    gpsjam.release_resident()


def test_resident_capture_eviction_keeps_held_captures_alive(tmp_path, monkeypatch):
    """Above GPSJAM_RESIDENT_GIB the cache lets go of the least recently used capture, but a caller that
    still holds it (a scan in another thread) keeps valid memory; it is freed with its last reference."""
    monkeypatch.setattr(gpsjam, "_default", None)
    gpsjam.release_resident()
    monkeypatch.setenv("GPSJAM_RESIDENT_GIB", str(300000 / (1 << 30)))          # room for one 200 kB file
    paths = []
    for k in range(2):
        p = tmp_path / f"e{k}.bin"
        np.full(200000, 10 + k, np.uint8).tofile(p)
        paths.append(str(p))
    a = gpsjam.resident_capture(paths[0])
    b = gpsjam.resident_capture(paths[1])                                      # evicts a from the cache
    assert len(gpsjam._resident) == 1 and a.ptr and b.ptr
    assert a.download(0, 4).tolist() == [10] * 4 and b.download(0, 4).tolist() == [11] * 4
    np.testing.assert_array_equal(a.dev.chunk_power(a), a.dev.chunk_power(np.full(200000, 10, np.uint8)))
    before = gpsjam.Capture.uploads
    assert gpsjam.resident_capture(paths[0]) is not a and gpsjam.Capture.uploads == before + 1
    gpsjam.release_resident()


# ----------------------------------------------------------------------------- TDOA slots
def test_slots_equal_separate_slices(dev):
    n, sl = 500000, 50000                                    # the reference's own slice size (L = 2^17)
    delays = (0, 6, -9)
    raws = [generate(StreamSpec(seed=61, antenna=a, delay=d, jam_start=260000, jam_end=1 << 40, jam_sigma=70.0), n)
            for a, d in enumerate(delays)]
    sb = dev.tdoa_slot_bytes(sl)
    assert sb == sharded.slot_bytes(sl) and sb % 256 == 0
    slots = dev.alloc(3 * sb)
    d_on = dev.alloc(32)
    onsets = []
    for a, r in enumerate(raws):
        with dev.capture(r) as cap:
            dev.onset_dev(cap, cap.nbytes, 200000, 1000, 50.0, d_on)
            dev.tdoa_slot_dev(cap, cap.nbytes, d_on, sl, slots.ptr + a * sb)
            dev.synchronize()
        onsets.append(int(d_on.download(np.int64, 1)[0]))
    host = slots.download(np.uint8)
    import torch
    for a, r in enumerate(raws):
        np.testing.assert_array_equal(host[a * sb:(a + 1) * sb], sharded.make_slot(torch.from_numpy(r), onsets[a], sl).numpy())
    pairs = [(0, 1), (0, 2), (1, 2), (2, 0)]
    d_l, d_p, d_m = dev.alloc(16), dev.alloc(16), dev.alloc(16)
    dev.xcorr_slots_dev(slots, sb, 3, sl, pairs, d_l, d_p, d_m)
    dev.synchronize()
    got = d_l.download(np.int32).tolist()
    sl_raw = [r[2 * o:2 * (o + sl)] for r, o in zip(raws, onsets)]
    want, peaks = dev.xcorr_lags(sl_raw, pairs)
    assert got == want.tolist()
    np.testing.assert_array_equal(d_p.download(np.float32), peaks)
    for (i, j), lag in zip(pairs, got):
        assert lag + onsets[j] - onsets[i] == delays[j] - delays[i]
    assert (d_m.download(np.float32) > 0.5).all()
    # bad stride / alignment are refused
    with pytest.raises(gpsjam.GpsJamError):
        dev.xcorr_slots_dev(slots, 2 * sl, 3, sl, pairs, d_l, d_p, d_m)


# ----------------------------------------------------------------------------- decision margins
def near_tie_capture(n=300000, burst=250000):
    """Quiet floor of constant power 2.5, then constant power 252.5: with 490 burst samples in the
    1000-sample window the exact moving average equals the exact threshold 125.0 (no crossing in
    exact arithmetic until the 491st); the reference's float32 |z|^2 rounds 252.5 up, so ITS
    average is above ITS threshold one index earlier."""
    raw = np.empty(2 * n, np.uint8)
    raw[0::2], raw[1::2] = 129, 127                          # z = (1.5, -0.5): |z|^2 = 2.5
    raw[2 * burst::2], raw[2 * burst + 1::2] = 143, 131      # z = (15.5, 3.5): |z|^2 = 252.5
    return raw


def test_k4_margins_and_near_tie_guard(dev, monkeypatch):
    import triangulateTDOA as tdoa
    monkeypatch.setattr(gpsjam, "default_device", lambda: dev)
    # (1) an ordinary capture: wide margins, no host work
    raw = generate(StreamSpec(seed=71, jam_start=230000, jam_end=1 << 40, jam_sigma=60.0), 400000)
    o = dev.onset(raw)
    assert o.start_index == orc.tdoa_onset(orc.tdoa_unpack(raw))
    assert o.margin_hit > 1e-5 and o.margin_before > 1e-5 and o.margin == min(o.margin_hit, o.margin_before)
    tdoa.near_tie_events.clear()
    assert tdoa.find_interference_start(tdoa.IQCapture(raw), 200000, 1000, 50.0) == o.start_index
    assert tdoa.near_tie_events == []
    # never crosses: the margin says how far below the threshold the loudest window stayed
    quiet = generate(StreamSpec(seed=72), 300000)
    q = dev.onset(quiet)
    assert q.start_index == -1 and 0.9 < q.margin_before < 1.0 and q.margin == q.margin_before
    # (2) the constructed near-tie: exact arithmetic and the reference differ by one index
    raw = near_tie_capture()
    want = orc.tdoa_onset(orc.tdoa_unpack(raw))
    assert want == 249990                                     # what the reference returns (probed in the build container)
    o = dev.onset(raw)
    assert o.start_index == 249991                            # the exact crossing
    assert abs(o.margin_before) < 1e-6 and o.margin < tdoa.ONSET_NEAR_TIE
    tdoa.near_tie_events.clear()
    assert tdoa.find_interference_start(tdoa.IQCapture(raw), 200000, 1000, 50.0) == want
    assert tdoa.near_tie_events and tdoa.near_tie_events[0][0] == "onset"
    # the fused scan reports the same margins as K4 alone
    buf = dev.alloc(raw.size).upload(raw)
    d_pow, d_amp, d_on = dev.alloc(4 * dev.chunk_count(raw.size, 65536)), dev.alloc(32), dev.alloc(32)
    dev.stream_scan_dev(buf, raw.size, 65536, d_pow, 0.0, d_amp, 200000, 1000, 50.0, d_on)
    dev.synchronize()
    f = _ffi.Onset.from_buffer_copy(d_on.download(np.uint8, 32).tobytes())
    assert f.start_index == o.start_index and abs(f.margin_before) < 1e-6
    assert f.margin_hit == pytest.approx(o.margin_hit, rel=1e-6)


def test_k5_margin_and_symmetric_tie(dev, monkeypatch):
    """Two slices that are mirror-symmetric about their centre correlate to EXACTLY equal
    magnitudes at lags +d and -d; which one an FFT implementation's arg-max returns is decided by
    its rounding.  The margin reports the tie and the drop-in lets the reference's own call decide."""
    import triangulateTDOA as tdoa
    from scipy import signal
    monkeypatch.setattr(gpsjam, "default_device", lambda: dev)
    n, m, d = 50000, 512, 1000
    rng = np.random.default_rng(5)
    half = rng.integers(20, 236, size=(m // 2, 2), dtype=np.uint8)
    burst = np.concatenate([half, half[::-1]])               # palindrome in I and in Q
    s0 = np.full((n, 2), 128, np.uint8)
    s1 = np.full((n, 2), 128, np.uint8)
    c = n // 2 - m // 2
    s0[c:c + m] = burst
    s1[c + d:c + d + m] = burst
    s1[c - d:c - d + m] = burst
    r0, r1 = s0.reshape(-1), s1.reshape(-1)
    lags, peaks, margins = dev.xcorr_lags([r0, r1], [(0, 1)], want_margins=True)
    assert abs(int(lags[0])) == d and margins[0] < tdoa.LAG_NEAR_TIE
    corr = signal.correlate(orc.tdoa_unpack(r1), orc.tdoa_unpack(r0), mode="full")   # the reference's call (:86)
    want = int(np.argmax(np.abs(corr))) - (n - 1)
    tdoa.near_tie_events.clear()
    lag, peak = tdoa.correlation_lag(tdoa.IQCapture(r1), tdoa.IQCapture(r0))
    assert lag == want and tdoa.near_tie_events and tdoa.near_tie_events[0][0] == "lag"
    assert lag == orc.xcorr_lag(orc.tdoa_unpack(r1), orc.tdoa_unpack(r0))[0]
    # a single clean peak: wide margin, GPU result stands
    s1b = np.full((n, 2), 128, np.uint8)
    s1b[c + d:c + d + m] = burst
    tdoa.near_tie_events.clear()
    lag, _ = tdoa.correlation_lag(tdoa.IQCapture(s1b.reshape(-1)), tdoa.IQCapture(r0))
    assert lag == d and tdoa.near_tie_events == []


# ----------------------------------------------------------------------------- RCCL communicator
def test_comm_single_rank_gather_and_bcast(dev):
    """gj_comm_* on a one-rank communicator: RCCL is loaded at run time, the communicator binds to
    the context's GPU, gather / broadcast move device buffers on the context's stream."""
    from gpsjam.comm import Communicator
    with Communicator(dev, rank=0, world_size=1) as comm:
        r, w = C.c_int(-1), C.c_int(-1)
        assert dev._lib.gj_comm_rank(comm._h, C.byref(r), C.byref(w)) == 0 and (r.value, w.value) == (0, 1)
        n = 1 << 20
        src = np.random.default_rng(3).integers(0, 256, n, dtype=np.uint8)
        d_src, d_dst = dev.alloc(n).upload(src), dev.alloc(n)
        comm.gather(d_src, n, d_dst, 0)
        comm.bcast(d_dst, n, 0)
        dev.synchronize()
        np.testing.assert_array_equal(d_dst.download(np.uint8), src)
        with pytest.raises(gpsjam.GpsJamError):
            comm.gather(d_src, n, d_dst, 3)                 # no such root


def test_antenna_stream_over_native_transport(dev):
    """The pipeline with transport = a gj_comm communicator (one rank): same results as the torch
    transport."""
    import torch
    from gpsjam.comm import Communicator
    n = 800000
    raw = generate(StreamSpec(seed=81, jam_start=350000, jam_end=1 << 40, jam_sigma=60.0), n)
    work = torch.cuda.Stream()
    torch.cuda.set_stream(work)
    d = gpsjam.Device(0)
    d.set_stream(work.cuda_stream)
    cap = torch.from_numpy(raw).cuda()
    a = sharded.AntennaStream(d, cap, nperseg=1024, chunk_samples=300000, slice_samples=1 << 15)
    va = a.step()[0].clone()
    b = sharded.AntennaStream(d, cap, nperseg=1024, chunk_samples=300000, slice_samples=1 << 15, overlap=False,
                              transport=Communicator(d, 0, 1))
    vb = b.step()[0].clone()
    torch.cuda.synchronize()
    assert torch.equal(va, vb)
    a.close()
    b.close()
    torch.cuda.set_stream(torch.cuda.default_stream())
    d.close()


# ----------------------------------------------------------------------------- unpack convention
def test_unpack_convention_offset_128(dev):
    """gj_set_unpack(128, 1/128): the gnssdec convention (sdrrcv.c:104-106) through K1..K5, against the
    same numpy / scipy expressions the oracle uses with 127.5 (written out here for 128)."""
    from scipy import signal
    n = 700000
    raw = generate(StreamSpec(seed=91, jam_start=300000, jam_end=1 << 40, jam_sigma=55.0, dc_i_q8=900), n)
    assert dev.get_unpack() == (127.5, 1.0 / 127.5)
    ref_default = dev.chunk_power(raw)
    try:
        dev.set_unpack(128.0, 1.0 / 128.0)
        assert dev.get_unpack() == (128.0, 1.0 / 128.0)
        i8 = raw.astype(np.int32) - 128
        # K1: exact integer mean of (I-128)^2 + (Q-128)^2 per 64-KiB chunk, rounded once
        pm = dev.chunk_power(raw, eps=0.0)
        want = []
        for o in range(0, raw.size, 65536):
            p = i8[o:o + 65536]
            want.append(np.float32(float((p[0::2] ** 2 + p[1::2] ** 2).sum()) / (p.size // 2)))
        np.testing.assert_array_equal(pm, np.array(want, np.float32))
        assert not np.array_equal(pm, ref_default)
        # K3: |x|, x = (u - 128)/128
        z = (i8[0::2] + 1j * i8[1::2]) / 128.0
        amp = np.abs(z)
        for thr in (0.0, 0.25):
            st = dev.amp_stats(raw, thr)
            k = int(np.argmax(amp > thr))
            assert st.first_index == k and st.count == n - k
            np.testing.assert_allclose(st.mean, amp[k:].mean(), rtol=1e-6)
        # K4 (LSB units, like its reference): exact integer window sums against the float64 expression
        zz = i8[0::2].astype(np.float64) + 1j * i8[1::2]
        pw = np.abs(zz) ** 2
        thr = np.float32(pw[:200000].mean()) * np.float32(50.0)
        cs = np.concatenate([[0.0], np.cumsum(pw)])
        ma = (cs[1000:] - cs[:-1000]) / 1000.0
        want_on = int(np.argmax(ma > thr)) + 500
        o = dev.onset(raw)
        assert o.margin > 1e-6 and o.start_index == want_on
        # K2: scipy.signal.welch on x = (u - 128)/128 with widmo_plot's chunk rule
        psd, _ = dev.welch(raw, chunk_samples=300000, nperseg=4096, want_db=False)
        x = z.astype(np.complex64)
        for c in range(psd.shape[0]):
            seg = x[c * 300000:(c + 1) * 300000]
            seg = seg - np.mean(seg)
            _, p = signal.welch(seg, 2.048e6, nperseg=4096, return_onesided=False)
            p = np.fft.fftshift(p)
            keep = p > 1e-12
            assert np.max(np.abs(psd[c][keep] - p[keep]) / p[keep]) < 1e-4
        # K5: a constant offset does not move the lag of a clean peak
        sl = [raw[2 * want_on:2 * (want_on + 32768)], raw[2 * (want_on - 7):2 * (want_on - 7 + 32768)]]
        lags, _ = dev.xcorr_lags(sl, [(0, 1)])
        assert lags.tolist() == [7]
        # the fused pass follows the same convention
        buf = dev.alloc(raw.size).upload(raw)
        d_pow, d_amp, d_on = dev.alloc(4 * dev.chunk_count(raw.size, 65536)), dev.alloc(32), dev.alloc(32)
        dev.stream_scan_dev(buf, raw.size, 65536, d_pow, 0.0, d_amp, 200000, 1000, 50.0, d_on, eps=0.0)
        dev.synchronize()
        np.testing.assert_array_equal(d_pow.download(np.float32), pm)
        a = _ffi.AmpStats.from_buffer_copy(d_amp.download(np.uint8, 32).tobytes())
        np.testing.assert_allclose(a.mean, amp.mean(), rtol=1e-6)
        assert int(d_on.download(np.int64, 1)[0]) == want_on
        with pytest.raises(gpsjam.GpsJamError):
            dev.set_unpack(127.3, 1.0)                        # not a multiple of 0.5
    finally:
        dev.set_unpack()                                      # back to the reference's convention
    np.testing.assert_array_equal(dev.chunk_power(raw), ref_default)


# ----------------------------------------------------------------------------- the binding shown in INTEGRATION.md
def test_integration_md_ctypes_stub_runs():
    """The minimal ctypes binding printed in INTEGRATION.md section B is executed as it stands (only the library path
    is made absolute) and gives the oracle's numbers."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    start = text.index("```python\n# gpsjam_ffi.py") + len("```python\n")
    code = text[start:text.index("```", start)]
    assert 'C.CDLL("csrc/libgpsjam_hip.so")' in code
    code = code.replace('"csrc/libgpsjam_hip.so"', repr(gpsjam.library_path()))
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    raw = generate(StreamSpec(seed=8, jam_start=60000, jam_end=1 << 40, jam_sigma=55.0), 150001)
    np.testing.assert_allclose(ns["chunk_power"](raw), orc.chunk_power(raw), rtol=1e-6)
    k, avg = ns["amp_stats"](raw, 0.5)
    want_k, want_avg = orc.rssi_amp_stats(raw, 0.5)
    assert k == want_k
    np.testing.assert_allclose(avg, want_avg, rtol=1e-6)
    assert ns["amp_stats"](raw, 5.0) == (None, None)
    ns["lib"].gj_destroy(ns["ctx"])
