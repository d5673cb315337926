"""Test double for gpsjam.Device backed by the CPU oracle.

ONLY for the ``-m "not gpu"`` tests of the host-side logic of the drop-in modules (result
dicts, Qt signal plumbing, detector state machine, CLI output): it lets that logic run in the
GPU-less build container.  It lives under tests/ and is never importable from the product.
"""
import ctypes as C

import numpy as np

from gpsjam import _ffi
from oracle import gpsjam_oracle as orc


class FakeCapture:
    """Stand-in for gpsjam.Capture: the bytes stay on the host."""

    def __init__(self, dev, source):
        import os
        self.dev = dev
        if isinstance(source, (str, bytes, os.PathLike)):
            self.raw = np.fromfile(source, dtype=np.uint8)
        else:
            self.raw = np.ascontiguousarray(np.asarray(source, dtype=np.uint8))
        self.nbytes, self.ptr = int(self.raw.size), 1
        self.results, self.ingest_ms = {}, None

    def free(self):
        self.ptr = 0


def _bytes_of(raw):
    return raw.raw if isinstance(raw, FakeCapture) else np.asarray(raw, dtype=np.uint8)


class OracleDevice:
    last_kernel_ms = 0.0

    def capture(self, source, offset=0, max_bytes=0):
        return FakeCapture(self, source)

    def ingest(self, source, **kw):
        return FakeCapture(self, source)          # the test double computes on demand

    def byte_histogram(self, cap, chunk_samples=2048000, nperseg=1024, stride=100):
        _, _, samples = orc.widmo_waterfall(_bytes_of(cap), nperseg=nperseg, chunk_samples=chunk_samples)
        return np.bincount(samples, minlength=256).astype(np.uint64)

    def chunk_power(self, raw, chunk_bytes=65536, eps=1e-10, odd_chunk_zero=False):
        raw = np.ascontiguousarray(_bytes_of(raw))
        if odd_chunk_zero:
            out = [orc.cij_chunk_power(raw[o:o + chunk_bytes], 0.0)[1] for o in range(0, raw.size, chunk_bytes)]
            return np.array(out, np.float32)
        return orc.chunk_power(raw, chunk_bytes).astype(np.float32)

    def amp_stats(self, raw, threshold):
        raw = _bytes_of(raw)
        k, avg = orc.rssi_amp_stats(raw, threshold)
        st = _ffi.AmpStats()
        st.first_index = -1 if k is None else k
        st.count = 0 if k is None else len(raw) // 2 - k
        st.mean = 0.0 if k is None else float(avg)
        st.sum = st.mean * st.count
        return st

    def onset(self, raw, noise_samples=200000, window=1000, factor=50.0):
        o = _ffi.Onset()
        o.start_index = orc.tdoa_onset(orc.tdoa_unpack(_bytes_of(raw)), noise_samples, window, factor)
        o.margin_hit = o.margin_before = 1.0
        return o

    def xcorr_lags(self, slices, pairs, want_margins=False):
        z = [orc.tdoa_unpack(np.asarray(s, dtype=np.uint8)) for s in slices]
        res = [orc.xcorr_lag(z[j], z[i]) for i, j in pairs]
        out = (np.array([r[0] for r in res], np.int32), np.array([r[1] for r in res], np.float32))
        return out + (np.ones(len(res), np.float32),) if want_margins else out

    def welch(self, raw, chunk_samples=2048000, nperseg=1024, fs=2.048e6, shift=True, want_db=True):
        lin, db, _ = orc.widmo_waterfall(_bytes_of(raw), fs, nperseg, chunk_samples)
        if not shift:
            lin, db = np.fft.ifftshift(lin, axes=1), np.fft.ifftshift(db, axes=1)
        return lin, (db if want_db else None)
