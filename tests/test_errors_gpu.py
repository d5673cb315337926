"""Error behaviour of the C-ABI on a live context: bad parameters come back as status codes with
a message (surfaced as GpsJamError by the ctypes layer), never as a crash, and the context stays
usable afterwards."""
import numpy as np
import pytest

import gpsjam
from gpsjam.synth import StreamSpec, generate

pytestmark = pytest.mark.gpu

GJ_ERR_INVALID, GJ_ERR_UNSUPPORTED = -1, -5


def test_bad_parameters_are_reported_and_context_survives(dev):
    raw = generate(StreamSpec(seed=3), 50000)
    with pytest.raises(gpsjam.GpsJamError) as e:
        dev.welch(raw, chunk_samples=20000, nperseg=1000)            # not a power of two
    assert e.value.status == GJ_ERR_UNSUPPORTED and "nperseg" in str(e.value)
    with pytest.raises(gpsjam.GpsJamError) as e:
        dev.welch(raw, chunk_samples=20000, nperseg=8192)            # above the supported size
    assert e.value.status == GJ_ERR_UNSUPPORTED
    with pytest.raises(gpsjam.GpsJamError) as e:
        dev.onset(raw, 1000, 9000, 50.0)                             # window > 8192
    assert e.value.status == GJ_ERR_UNSUPPORTED and "window" in str(e.value)
    with pytest.raises(gpsjam.GpsJamError) as e:
        dev.onset(raw, 0, 1000, 50.0)
    assert e.value.status == GJ_ERR_INVALID
    with pytest.raises(gpsjam.GpsJamError) as e:
        dev.chunk_power(raw, chunk_bytes=0)
    assert e.value.status == GJ_ERR_INVALID
    with pytest.raises(gpsjam.GpsJamError) as e:
        dev.xcorr_lags([raw] * 17, [(0, 1)])                         # more than GJ_MAX_ANTENNAS
    assert e.value.status == GJ_ERR_INVALID
    d_psd = dev.alloc(4 * 4096)
    with pytest.raises(gpsjam.GpsJamError) as e:                     # chunk shorter than one segment
        dev.welch_dev(dev.alloc(raw.size).upload(raw), raw.size, 1000, 4096, 2.048e6, d_psd)
    assert e.value.status == GJ_ERR_UNSUPPORTED
    with pytest.raises(gpsjam.GpsJamError) as e:                     # one chunk = 2^31 samples: 32-bit chunk offsets
        dev.welch_dev(dev.alloc(raw.size).upload(raw), raw.size, 1 << 31, 4096, 2.048e6, d_psd)
    assert e.value.status == GJ_ERR_UNSUPPORTED
    # and the context still works
    pm = dev.chunk_power(raw)
    assert pm.shape == (2,) and np.isfinite(pm).all()


def test_empty_and_tiny_inputs(dev):
    assert dev.chunk_power(np.zeros(0, np.uint8)).size == 0
    psd, _ = dev.welch(np.zeros(10, np.uint8), nperseg=1024)
    assert psd.shape == (0, 1024)
    st = dev.amp_stats(np.zeros(0, np.uint8), 0.0)
    assert st.first_index == -1 and st.count == 0
    assert dev.onset(np.zeros(100, np.uint8), 200000, 1000, 50.0).start_index == -1


def test_capture_part_arguments_are_checked(dev):
    """gj_part_*: a view that does not satisfy the alignment rules of include/gpsjam.h is refused with a status,
    never run (a part whose own range starts off a chunk boundary, a halo that is not whole tiles or too short for the
    window, a part without the capture's noise span, an own range outside the buffer)."""
    import ctypes as C
    from gpsjam import _ffi
    n = 40 * 65536
    raw = np.full(n, 128, np.uint8)
    buf = dev.alloc(n).upload(raw)
    d_pow, d_tiles, d_amp, d_on = dev.alloc(4 * 64), dev.alloc(16 * 64), dev.alloc(32), dev.alloc(32)

    def scan(view, window=1000):
        return dev._lib.gj_part_scan_dev(dev._ctx, C.byref(view), 65536, 1e-10, 0, d_pow.ptr, 0.0, d_tiles.ptr, d_amp.ptr,
                                         20000, window, 50.0, d_on.ptr)

    total = 100 * 65536
    ok = _ffi.PartView(buf.ptr, n, 9 * 65536, 10 * 65536, 20 * 65536, total, buf.ptr)
    assert scan(ok) == 0
    dev.synchronize()
    bad_start = _ffi.PartView(buf.ptr, n, 9 * 65536, 10 * 65536 + 4096, 20 * 65536, total, buf.ptr)
    assert scan(bad_start) == -5                                                 # GJ_ERR_UNSUPPORTED
    bad_halo = _ffi.PartView(buf.ptr, n, 10 * 65536 - 4096, 10 * 65536, 20 * 65536, total, buf.ptr)
    assert scan(bad_halo) == -5
    short_halo = _ffi.PartView(buf.ptr, n, 9 * 65536, 10 * 65536, 20 * 65536, total, buf.ptr)
    assert dev._lib.gj_part_scan_dev(dev._ctx, C.byref(short_halo), 65536, 1e-10, 0, d_pow.ptr, 0.0, d_tiles.ptr, d_amp.ptr,
                                     20000, 8192, 50.0, d_on.ptr) == 0          # 32768 samples of halo hold the largest window
    no_noise = _ffi.PartView(buf.ptr, n, 9 * 65536, 10 * 65536, 20 * 65536, total, None)
    assert scan(no_noise) == -1                                                  # GJ_ERR_INVALID
    assert b"noise span" in dev._lib.gj_last_error(dev._ctx)
    outside = _ffi.PartView(buf.ptr, n, 9 * 65536, 10 * 65536, 60 * 65536, total, buf.ptr)
    assert scan(outside) == -1
    ragged_middle = _ffi.PartView(buf.ptr, n, 9 * 65536, 10 * 65536, 20 * 65536 + 100, total, buf.ptr)
    assert scan(ragged_middle) == -5                                             # only the capture's last part may end ragged
    d_psd = dev.alloc(4 * 1024 * 64)
    assert dev._lib.gj_part_welch_dev(dev._ctx, C.byref(bad_start), 32768, 1024, 2.048e6, 1, d_psd.ptr, None) == -5
    assert dev._lib.gj_part_welch_dev(dev._ctx, C.byref(ok), 32768, 1024, 2.048e6, 1, d_psd.ptr, None) == 0
    dev.synchronize()
    pm = dev.chunk_power(raw[:4 * 65536])                                        # the context still works
    assert pm.shape == (4,)
    for b in (buf, d_pow, d_tiles, d_amp, d_on, d_psd):
        b.free()


def test_round5_entry_points_check_their_arguments(dev):
    """gj_capture_scan_dev, gj_welch_batch_dev, gj_pack_results_dev, gj_set_fill_threads, gj_debug_inject, gj_probe_busy_dev:
    bad arguments come back as a status, nothing is launched, and the context keeps working."""
    from gpsjam import _ffi
    n = 400_000
    raw = generate(StreamSpec(seed=8, jam_start=250_000, jam_end=1 << 40, jam_sigma=50.0), n)
    cap = dev.alloc(2 * n + 16).upload(raw)
    nch = dev.chunk_count(2 * n, 65536)
    d_pow, d_st, d_amp, d_on = dev.alloc(4 * nch), dev.alloc(16), dev.alloc(32), dev.alloc(32)
    d_slot = dev.alloc(dev.tdoa_slot_bytes(4096) + 16)
    with pytest.raises(gpsjam.GpsJamError) as e:                     # a slot that is not 16-byte aligned
        dev.capture_scan_dev(cap, 2 * n, 65536, d_pow, 0.0, d_amp, 200000, 1000, 50.0, d_on, slice_samples=4096, d_slot=d_slot.ptr + 8)
    assert e.value.status == GJ_ERR_INVALID and "aligned" in str(e.value)
    with pytest.raises(gpsjam.GpsJamError) as e:                     # a slot of no samples
        dev.capture_scan_dev(cap, 2 * n, 65536, d_pow, 0.0, d_amp, 200000, 1000, 50.0, d_on, slice_samples=0, d_slot=d_slot)
    assert e.value.status == GJ_ERR_INVALID
    with pytest.raises(gpsjam.GpsJamError) as e:
        dev.capture_scan_dev(cap, 2 * n, 65536, d_pow, 0.0, d_amp, 200000, 9000, 50.0, d_on)       # window > 8192
    assert e.value.status == GJ_ERR_UNSUPPORTED
    with pytest.raises(gpsjam.GpsJamError) as e:
        dev.capture_scan_dev(cap, 2 * n, 0, d_pow, 0.0, d_amp, 200000, 1000, 50.0, d_on)
    assert e.value.status == GJ_ERR_INVALID
    rows = dev.welch_rows(2 * n, 100_000, 1024)
    psd = dev.alloc(4 * rows * 1024 + 16)
    with pytest.raises(gpsjam.GpsJamError) as e:                     # PSD rows off a 16-byte boundary
        dev.welch_batch_dev([cap, cap], 2 * n, 100_000, 1024, 2.048e6, [psd, psd.ptr + 4])
    assert e.value.status == GJ_ERR_INVALID and "aligned" in str(e.value)
    with pytest.raises(gpsjam.GpsJamError) as e:
        dev.welch_batch_dev([cap] * 17, 2 * n, 100_000, 1024, 2.048e6, [psd] * 17)
    assert e.value.status == GJ_ERR_INVALID
    with pytest.raises(gpsjam.GpsJamError) as e:
        dev.welch_batch_dev([cap, cap], 2 * n, 100_000, 1000, 2.048e6, [psd, psd])
    assert e.value.status == GJ_ERR_UNSUPPORTED
    desc = _ffi.CombineCapture(n_chunks=nch, rows=rows, n_tiles=0, total_bytes=0, n_parts=1, antenna=0, n_pairs=2, pair_cap=1,
                               d_power=d_pow.ptr, d_stats=d_st.ptr, d_tiles=None, d_amp_parts=None, d_onset_parts=None, d_amp=d_amp.ptr,
                               d_onset=d_on.ptr, d_psd=psd.ptr, d_out=psd.ptr)
    with pytest.raises(gpsjam.GpsJamError) as e:                     # more pairs than capacity
        dev.pack_results_dev([desc], 1024)
    assert e.value.status == GJ_ERR_INVALID
    desc.n_pairs, desc.d_out = 0, None
    with pytest.raises(gpsjam.GpsJamError) as e:                     # no output vector
        dev.pack_results_dev([desc], 1024)
    assert e.value.status == GJ_ERR_INVALID
    for call in (lambda: dev.set_fill_threads(99), lambda: dev.debug_inject(7, 1), lambda: dev.probe_busy_dev(-1.0)):
        with pytest.raises(gpsjam.GpsJamError) as e:
            call()
        assert e.value.status == GJ_ERR_INVALID
    # and the context still works: the scan that was refused four times now runs
    dev.capture_scan_dev(cap, 2 * n, 65536, d_pow, 0.0, d_amp, 200000, 1000, 50.0, d_on, d_stats=d_st, slice_samples=4096, d_slot=d_slot)
    dev.synchronize()
    assert int(d_on.download(np.int64, 1)[0]) > 0 and np.isfinite(d_pow.download(np.float32, nch)).all()
    for b in (cap, d_pow, d_st, d_amp, d_on, d_slot, psd):
        b.free()
