"""Round-3 additions on the GPU: the fused scan with an integer unpack offset, decision margins in the
result vector, the bounded near-tie re-evaluation, lock-free waits, ingest overlapped with compute."""
import ctypes as C
import os
import sys
import threading
import time

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


def _fused(dev, raw, thr):
    buf = dev.alloc(max(raw.size, 16)).upload(raw)
    d_pow, d_amp, d_on = dev.alloc(4 * max(dev.chunk_count(raw.size, 65536), 1)), dev.alloc(32), dev.alloc(32)
    dev.stream_scan_dev(buf, raw.size, 65536, d_pow, thr, d_amp, 200000, 1000, 50.0, d_on)
    dev.synchronize()
    a = _ffi.AmpStats.from_buffer_copy(d_amp.download(np.uint8, 32).tobytes())
    for b in (buf, d_pow, d_amp, d_on):
        b.free()
    return a


# ----------------------------------------------------------------------------- fused scan, integer offset
@pytest.mark.parametrize("thr", [0.0, 0.004, 0.02])
def test_fused_scan_integer_offset_tracks_first_index(dev, thr):
    """With gj_set_unpack(128, 1/128) an amplitude can be exactly zero: a capture that starts with 0x80,0x80 samples
    has its first sample above a zero threshold where the bytes first move.  The fused pass must agree with the
    stand-alone K3 (`a > thr`) -- it used to assume every amplitude is >= sqrt(2)/255 and report first = 0."""
    n = 400000
    raw = generate(StreamSpec(seed=17, jam_start=250000, jam_end=1 << 40, jam_sigma=50.0), n).copy()
    lead = 70001                                        # samples of exact zeros in the 128 convention, ends mid-tile
    raw[:2 * lead] = 0x80
    try:
        dev.set_unpack(128.0, 1.0 / 128.0)
        alone = dev.amp_stats(raw, thr)
        fused = _fused(dev, raw, thr)
        i8 = raw.astype(np.int32) - 128
        amp = np.abs((i8[0::2] + 1j * i8[1::2]) / 128.0)
        k = int(np.argmax(amp > thr))
        assert k >= lead
        assert (alone.first_index, alone.count) == (k, n - k)
        assert (fused.first_index, fused.count) == (alone.first_index, alone.count)
        assert fused.sum == alone.sum and fused.mean == alone.mean
        np.testing.assert_allclose(fused.mean, amp[k:].mean(), rtol=1e-6)
    finally:
        dev.set_unpack()
    # the default convention keeps its shortcut and its answer: every sample counts from index 0
    d = _fused(dev, raw, 0.0)
    assert (d.first_index, d.count) == (0, n)
    d2 = dev.amp_stats(raw, 0.0)
    assert (d2.first_index, d2.count, d2.sum) == (d.first_index, d.count, d.sum)


def test_fused_scan_small_scale_threshold_between(dev):
    """A small scale puts real amplitudes below the old 0.005 cut-off: the fused pass must still compare them."""
    n = 300000
    raw = generate(StreamSpec(seed=23, jam_start=200000, jam_end=1 << 40, jam_sigma=50.0), n)
    try:
        dev.set_unpack(127.5, 1.0 / 4096.0)             # quiet floor ~ 6 LSB -> amplitude ~ 0.002, burst ~ 0.017
        thr = 0.004
        alone = dev.amp_stats(raw, thr)
        fused = _fused(dev, raw, thr)
        assert 0 < alone.first_index < n
        assert (fused.first_index, fused.count, fused.sum) == (alone.first_index, alone.count, alone.sum)
    finally:
        dev.set_unpack()
