"""gpsjam -- host side of the MI355X jamming-detection DSP path.

``Device`` owns one ``gj_ctx`` (one GPU) and exposes the C-ABI of
``libgpsjam_hip.so`` in two flavours:

* host arrays in, numpy results out (``chunk_power``, ``welch``, ``amp_stats``,
  ``onset``, ``xcorr_lags``) -- what the drop-in modules under ``skrypty/`` and
  ``GpsJammerApp/app/`` call;
* device pointers in / out (``*_dev``) on a caller-supplied HIP stream -- what
  ``bench.py`` and the one-capture-per-GPU driver (``gpsjam.sharded``) call with torch
  tensors.

Nothing in this package computes on the CPU: without the HIP library every call
raises ``GpsJamLibraryError``.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _ffi
from ._ffi import (AmpStats, GpsJamError, GpsJamLibraryError, Onset, SynthParams,  # noqa: F401
                   GJ_LAG_INVALID, GJ_MAX_ANTENNAS)

__all__ = ["Device", "DevBuf", "Capture", "GpsJamError", "GpsJamLibraryError", "device_count",
           "library_path", "as_u8", "default_device", "read_capture", "resident_capture",
           "release_resident"]

_default = None
_default_lock = __import__("threading").Lock()


def default_device() -> "Device":
    """Process-wide context used by the drop-in modules (GPU index from GPSJAM_DEVICE,
    default 0).  Raises if the HIP library or the GPU is missing -- there is no CPU path."""
    global _default
    with _default_lock:
        if _default is None or _default._ctx is None:
            _default = Device(int(__import__("os").environ.get("GPSJAM_DEVICE", "0")))
        return _default


def read_capture(path) -> np.ndarray:
    """uint8 view of a capture file without copying it through Python (memory-mapped;
    empty files give an empty array)."""
    import os
    if os.path.getsize(path) == 0:
        return np.zeros(0, np.uint8)
    return np.memmap(path, dtype=np.uint8, mode="r")


_resident = {}            # (realpath, size, mtime_ns) -> Capture, insertion order = LRU order
_resident_loading = {}    # key -> (threading.Event, kernel thread id of the loading thread): an upload in flight
_resident_lock = __import__("threading").Lock()


def resident_capture(path, **ingest) -> "Capture":
    """The capture file ``path`` in HBM on the default device, uploaded on first use and kept
    (process-wide, least-recently-used eviction above GPSJAM_RESIDENT_GIB, default 64 of the
    288 GB; an evicted capture is freed once nobody holds it any more): the worker's power scan, the RSSI solver and the PSD script all read the same
    files (worker.py:209-217 -> :590-600 -> triangulateRSSI.py:29), and each used to pay its own
    host->device pass.  ``ingest`` (keyword arguments of ``Device.ingest``): what the caller is about to
    compute -- on the first use of a file it is computed WHILE the file uploads and rides on the Capture; a
    file that is already resident is returned as it is (the caller's call then runs on it as usual).
    Raises FileNotFoundError like open().

    The cache lock is held for dictionary operations only, never across an upload: two threads bringing in
    different files do not wait for each other, and a thread that is stopped inside its upload (the GUI ends an
    analysis with QThread.terminate(), ui_mainwindow.py:818-826) leaves nothing locked -- a second caller of the
    same file waits for the upload in flight only while the thread that started it is alive, then takes over."""
    import os
    import threading
    st = os.stat(path)
    key = (os.path.realpath(path), st.st_size, st.st_mtime_ns)
    dev = default_device()
    while True:
        with _resident_lock:
            cap = _resident.pop(key, None)
            if cap is not None and cap.ptr and cap.dev is dev:
                _resident[key] = cap                      # most recently used last
                return cap
            pending = _resident_loading.get(key)
            if pending is not None and not _thread_alive(pending[1]):
                pending[0].set()                          # its loader is gone: whoever waits may try again
                del _resident_loading[key]
                pending = None
            if pending is None:
                mine = threading.Event()
                _resident_loading[key] = (mine, threading.get_native_id())
                limit = int(float(os.environ.get("GPSJAM_RESIDENT_GIB", "64")) * (1 << 30))
                while _resident and sum(c.nbytes for c in _resident.values()) + st.st_size > limit:
                    # eviction only drops the cache's reference: a caller that still holds the Capture (a scan
                    # running in another thread) keeps it alive, and its memory is freed when that reference goes
                    _resident.pop(next(iter(_resident)))
                break
        pending[0].wait(0.05)                             # short waits: the loader's liveness is re-checked above
    try:
        cap = dev.ingest(path, **ingest) if ingest else dev.capture(path)
        with _resident_lock:
            _resident[key] = cap
        return cap
    finally:
        with _resident_lock:
            if _resident_loading.get(key, (None,))[0] is mine:
                del _resident_loading[key]
        mine.set()


def _thread_alive(native_id) -> bool:
    """Kernel thread ``native_id`` of this process still exists (works for threads Python did not start, such as
    a QThread: the threading module does not know when those end)."""
    import os
    return os.path.exists(f"/proc/self/task/{native_id}")


def release_resident():
    """Free every cached capture (tests; long-running hosts that switch file sets)."""
    with _resident_lock:
        caps = list(_resident.values())
        _resident.clear()
    for c in caps:
        c.free()


def library_path() -> str:
    return _ffi.LIB_PATH


def device_count() -> int:
    n = C.c_int(0)
    _ffi.load().gj_device_count(C.byref(n))
    return n.value


def as_u8(raw) -> np.ndarray:
    """Contiguous uint8 view of bytes / bytearray / memoryview / ndarray."""
    if isinstance(raw, np.ndarray):
        if raw.dtype != np.uint8:
            raise TypeError("raw I/Q must be uint8")
        return np.ascontiguousarray(raw).reshape(-1)
    return np.frombuffer(raw, dtype=np.uint8)


def _ptr(x) -> int:
    """Device address of a torch tensor / DevBuf / int."""
    if x is None:
        return 0
    if isinstance(x, int):
        return x
    if isinstance(x, (DevBuf, Capture)):
        return x.ptr
    if hasattr(x, "data_ptr"):
        return int(x.data_ptr())
    raise TypeError(f"not a device buffer: {type(x)!r}")


class DevBuf:
    """Device allocation owned by a Device (for callers without torch)."""

    def __init__(self, dev: "Device", nbytes: int):
        self.dev, self.nbytes = dev, int(nbytes)
        p = C.c_void_p()
        dev._check(dev._lib.gj_malloc(dev._ctx, self.nbytes, C.byref(p)))
        self.ptr = p.value or 0

    def upload(self, host: np.ndarray, offset: int = 0):
        host = np.ascontiguousarray(host)
        if offset + host.nbytes > self.nbytes:
            raise ValueError("upload exceeds the device buffer")
        self.dev._check(self.dev._lib.gj_memcpy_h2d(self.dev._ctx, self.ptr + offset,
                                                    host.ctypes.data, host.nbytes))
        return self

    def download(self, dtype=np.uint8, count: Optional[int] = None, offset: int = 0) -> np.ndarray:
        dt = np.dtype(dtype)
        if count is None:
            count = (self.nbytes - offset) // dt.itemsize
        out = np.empty(count, dt)
        self.dev._check(self.dev._lib.gj_memcpy_d2h(self.dev._ctx, out.ctypes.data,
                                                    self.ptr + offset, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            self.dev._lib.gj_free(self.dev._ctx, self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Capture:
    """One capture resident in HBM: uploaded ONCE (file -> pinned bounce buffers -> HBM, or from
    a uint8 array), then handed to ``Device.chunk_power / welch / amp_stats / onset /
    byte_histogram / xcorr_lags_at`` any number of times.  The array-taking forms of those calls
    stage their input on every call (21 ms per GiB of PCIe against 0.2-1.3 ms of kernel time), so
    a caller that runs scan + PSD + RSSI on one file (widmo_plot, the worker's scan ->
    triangulation flow) uses this instead.  The library's ``*_u8`` entry points take the device
    address as it is, so a call on a Capture works in a lane of its own like any other: several
    host threads may use one Capture (and one Device) at once.  ``uploads`` counts host->device
    passes (tests)."""

    uploads = 0
    _count_lock = __import__("threading").Lock()

    @classmethod
    def _count(cls):
        with cls._count_lock:                                # several host threads may upload at once (one lane each)
            cls.uploads += 1

    @classmethod
    def _adopt(cls, dev: "Device", ptr: int, nbytes: int, path=None) -> "Capture":
        """A Capture around a device pointer the library has just handed out (gj_ingest_*)."""
        self = cls.__new__(cls)
        self.dev, self.ptr, self.nbytes, self.path = dev, int(ptr or 0), int(nbytes), path
        self.results, self.ingest_ms, self.results_unpack = {}, None, None
        Capture._count()
        return self

    def __init__(self, dev: "Device", source, offset: int = 0, max_bytes: int = 0):
        import os
        self.dev = dev
        # results computed while the capture was uploaded (Device.ingest), keyed by the call that would recompute
        # them: ("chunk_power", chunk_bytes, eps, odd_chunk_zero), ("amp_stats", threshold), ("onset", noise_samples,
        # window, factor), ("welch", chunk_samples, nperseg, fs, shift)
        self.results, self.ingest_ms, self.results_unpack = {}, None, None
        p, n = C.c_void_p(), C.c_size_t(0)
        if isinstance(source, (str, bytes, os.PathLike)):
            self.path = os.fspath(source)
            dev._check(dev._lib.gj_upload_file(dev._ctx, os.fsencode(self.path), int(offset), int(max_bytes),
                                               C.byref(p), C.byref(n)))
            self.nbytes = n.value
        else:
            self.path = None
            raw = as_u8(source)
            if offset or max_bytes:
                raw = raw[offset:offset + max_bytes] if max_bytes else raw[offset:]
            dev._check(dev._lib.gj_upload(dev._ctx, raw.ctypes.data if raw.size else None, raw.size, C.byref(p)))
            self.nbytes = int(raw.size)
        self.ptr = p.value or 0
        Capture._count()

    @property
    def nsamples(self) -> int:
        return self.nbytes // 2

    def download(self, offset: int = 0, count: Optional[int] = None) -> np.ndarray:
        """Bytes [offset, offset + count) back on the host (tests, near-tie re-evaluation)."""
        if count is None:
            count = self.nbytes - offset
        count = max(0, min(count, self.nbytes - offset))
        out = np.empty(count, np.uint8)
        if count:
            self.dev._check(self.dev._lib.gj_memcpy_d2h(self.dev._ctx, out.ctypes.data, self.ptr + offset, count))
        return out

    def free(self):
        if getattr(self, "ptr", 0) and getattr(self.dev, "_ctx", None):
            self.dev._lib.gj_free(self.dev._ctx, self.ptr)
        self.ptr = 0

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.free()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Device:
    def __init__(self, index: int = 0):
        self._lib = _ffi.load()
        ctx = C.c_void_p()
        rc = self._lib.gj_create(int(index), C.byref(ctx))
        if rc != 0:
            raise GpsJamError(rc, f"gj_create({index}): " + self._lib.gj_strerror(rc).decode())
        self._ctx = ctx
        self.index = int(index)
        self.last_kernel_ms = 0.0
        self.cache_hits = 0        # calls answered from a Capture's ride-along results (no kernel ran)
        self.kernel_calls = {}     # name -> host-array / resident-capture calls that DID reach the library

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc: int):
        if rc != 0:
            detail = self._lib.gj_last_error(self._ctx).decode(errors="replace")
            raise GpsJamError(rc, f"{self._lib.gj_strerror(rc).decode()}: {detail}")

    def capture(self, source, offset: int = 0, max_bytes: int = 0) -> Capture:
        """Upload a capture file (path) or a uint8 array once; see ``Capture``."""
        return Capture(self, source, offset, max_bytes)

    def ingest(self, source, *, chunk_bytes: int = 65536, eps: float = 1e-10, odd_chunk_zero: bool = False,
               rssi_threshold: float = 0.0, noise_samples: int = 200000, window: int = 1000, factor: float = 50.0,
               welch=None, fs: float = 2.048e6, shift: bool = True, want_db: bool = False, offset: int = 0,
               max_bytes: int = 0) -> Capture:
        """Upload a capture file (path) or uint8 array AND analyse it while it uploads (gj_ingest_*: the kernels run
        on the pieces that have landed: 1-16 MiB each, by capture size).  ``chunk_bytes`` != 0: the fused scan (K1 power map, K3 amplitude
        statistics at ``rssi_threshold``, K4 onset); ``welch=(chunk_samples, nperseg)``: the PSD waterfall.  The
        results ride on the returned Capture (``.results``) and are handed out by ``chunk_power`` / ``amp_stats`` /
        ``onset`` / ``welch`` when these are called on it with the same parameters -- bit-identical to what those
        calls compute on an uploaded capture, without a second pass."""
        import os
        kw = dict(chunk_bytes=chunk_bytes, eps=eps, odd_chunk_zero=odd_chunk_zero, rssi_threshold=rssi_threshold,
                  noise_samples=noise_samples, window=window, factor=factor, welch=welch, fs=fs, shift=shift, want_db=want_db)
        plan = self._ingest_plan(**kw)
        is_path = isinstance(source, (str, bytes, os.PathLike))
        if is_path:
            path = os.fspath(source)
            size = os.stat(path).st_size                     # FileNotFoundError like open()
            nbytes = max(0, size - int(offset))
            if max_bytes:
                nbytes = min(nbytes, int(max_bytes))
        else:
            path = None
            raw = as_u8(source)
            if offset or max_bytes:
                raw = raw[offset:offset + max_bytes] if max_bytes else raw[offset:]
            nbytes = int(raw.size)
        power, psd, db = self._ingest_buffers(nbytes, chunk_bytes, welch, want_db)
        res, p = _ffi.IngestResult(), C.c_void_p()
        args = (C.byref(plan), power.ctypes.data, power.size, psd.ctypes.data, db.ctypes.data if db is not None else None,
                psd.size, C.byref(res), C.byref(p))
        if is_path:
            self._check(self._lib.gj_ingest_file(self._ctx, os.fsencode(path), int(offset), int(max_bytes), *args))
        else:
            self._check(self._lib.gj_ingest_u8(self._ctx, raw.ctypes.data if raw.size else None, raw.size, *args))
        return self._ingest_adopt(p.value, res, path, power, psd, db, **kw)

    def ingest_many(self, paths, *, chunk_bytes: int = 65536, eps: float = 1e-10, odd_chunk_zero: bool = False,
                    rssi_threshold: float = 0.0, noise_samples: int = 200000, window: int = 1000, factor: float = 50.0,
                    welch=None, fs: float = 2.048e6, shift: bool = True, want_db: bool = False):
        """``ingest`` for several capture FILES at once (gj_ingest_files: one host thread of the library and one lane per
        file, the copies sharing the cores) -- the recordings of a deployment.  Returns the Captures in the order of
        ``paths``, each with the same ride-along results ``ingest`` would have left on it.  A file that cannot be read
        raises (FileNotFoundError from the stat here, GpsJamError from the library); captures that did come in are freed."""
        import os
        paths = [os.fspath(p) for p in paths]
        kw = dict(chunk_bytes=chunk_bytes, eps=eps, odd_chunk_zero=odd_chunk_zero, rssi_threshold=rssi_threshold,
                  noise_samples=noise_samples, window=window, factor=factor, welch=welch, fs=fs, shift=shift, want_db=want_db)
        if not paths:
            return []
        plan = self._ingest_plan(**kw)
        jobs = (_ffi.IngestJob * len(paths))()
        bufs = []
        for k, path in enumerate(paths):
            nbytes = os.stat(path).st_size                   # FileNotFoundError like open()
            power, psd, db = self._ingest_buffers(nbytes, chunk_bytes, welch, want_db)
            bufs.append((power, psd, db))
            j = jobs[k]
            j.path, j.offset, j.max_bytes = os.fsencode(path), 0, 0
            j.power, j.power_cap = power.ctypes.data, power.size
            j.psd, j.psd_db, j.psd_cap_floats = psd.ctypes.data, (db.ctypes.data if db is not None else None), psd.size
        rc = self._lib.gj_ingest_files(self._ctx, jobs, len(paths), C.byref(plan))
        if rc:
            for j in jobs:
                if j.status == 0 and j.dptr:
                    self._lib.gj_free(self._ctx, j.dptr)
            self._check(rc)
        return [self._ingest_adopt(jobs[k].dptr, jobs[k].result, paths[k], *bufs[k], **kw) for k in range(len(paths))]

    @staticmethod
    def _ingest_plan(*, chunk_bytes, eps, odd_chunk_zero, rssi_threshold, noise_samples, window, factor, welch, fs, shift, want_db):
        return _ffi.IngestPlan(int(chunk_bytes), float(eps), _ffi.GJ_CP_ODD_CHUNK_ZERO if odd_chunk_zero else 0,
                               float(rssi_threshold), int(noise_samples), int(window), float(factor),
                               int(welch[0]) if welch else 0, int(welch[1]) if welch else 0,
                               _ffi.GJ_WELCH_SHIFT if shift else 0, float(fs))

    def _ingest_buffers(self, nbytes, chunk_bytes, welch, want_db):
        n_chunks = self._lib.gj_chunk_count(nbytes, chunk_bytes) if chunk_bytes else 0
        rows = self._lib.gj_welch_rows(nbytes, welch[0], welch[1]) if welch else 0
        nper = int(welch[1]) if welch else 0
        power = np.empty(n_chunks, np.float32)
        psd = np.empty((rows, nper), np.float32)
        db = np.empty((rows, nper), np.float32) if (welch and want_db) else None
        return power, psd, db

    def _ingest_adopt(self, ptr, res, path, power, psd, db, *, chunk_bytes, eps, odd_chunk_zero, rssi_threshold, noise_samples,
                      window, factor, welch, fs, shift, want_db):
        cap = Capture._adopt(self, ptr, res.nbytes, path)
        cap.ingest_ms = (float(res.upload_ms), float(res.total_ms))
        cap.results_unpack = self.get_unpack()               # the convention the ride-along results were computed under
        for a in (power, psd, db):
            if a is not None:
                a.flags.writeable = False                    # handed out as they are on every matching call
        if chunk_bytes and power.size:
            cap.results[("chunk_power", int(chunk_bytes), float(np.float32(eps)), bool(odd_chunk_zero))] = power
            cap.results[("amp_stats", float(np.float32(rssi_threshold)))] = AmpStats.from_buffer_copy(bytes(res.amp))
            cap.results[("onset", int(noise_samples), int(window), float(np.float32(factor)))] = Onset.from_buffer_copy(bytes(res.onset))
        if welch and psd.shape[0]:
            cap.results[("welch", int(welch[0]), int(welch[1]), float(fs), bool(shift))] = (psd, db)
        return cap

    def _cached(self, raw, key):
        """Ride-along result of ``Device.ingest`` for this call, or None.  The results were computed under the unpack
        convention in force at ingest time: after a ``set_unpack`` to anything else they are dropped (gj_set_unpack
        promises to affect every later call on the context), and the call recomputes.  A hit ran no kernel:
        ``last_kernel_ms`` says so."""
        if not isinstance(raw, Capture) or not raw.results:
            return None
        if raw.results_unpack is not None and raw.results_unpack != self.get_unpack():
            if raw.dev is self:                              # this context's convention has changed: they are stale for good
                raw.results.clear()
            return None                                      # (another context with another convention: just not served)
        hit = raw.results.get(key)
        if hit is not None:
            self.last_kernel_ms = 0.0
            self.cache_hits += 1
        return hit

    def _count(self, name):
        self.kernel_calls[name] = self.kernel_calls.get(name, 0) + 1

    @staticmethod
    def _input(raw):
        """(address, nbytes, keep-alive) of a call's input: a resident Capture goes in by its device
        address (the library uses it in place), anything else as a host uint8 buffer (staged)."""
        if isinstance(raw, Capture):
            if not raw.ptr and raw.nbytes:
                raise ValueError("the capture has been freed")
            return raw.ptr, raw.nbytes, raw
        arr = as_u8(raw)
        return (arr.ctypes.data if arr.size else None), int(arr.size), arr

    def probe_busy_dev(self, milliseconds: float):
        """Keep this context's stream (and its hardware queue) busy for a while with one spinning wave (gj_probe_busy_dev):
        the stream-overlap probe of gpsjam/streams.py."""
        self._check(self._lib.gj_probe_busy_dev(self._ctx, float(milliseconds)))

    def debug_inject(self, what: int, count: int = 1):
        """Fault injection for tests (gj_debug_inject; GJ_INJECT_OWNER_ALIVE = 1)."""
        self._check(self._lib.gj_debug_inject(self._ctx, int(what), int(count)))

    def debug_counters(self):
        """Lanes made / in use / taken back from dead callers, dead-owner recoveries of the mutex."""
        v = [C.c_int(0) for _ in range(4)]
        self._check(self._lib.gj_debug_counters(self._ctx, *[C.byref(x) for x in v]))
        return dict(zip(("lanes", "lanes_busy", "lanes_reclaimed", "owner_deaths"), (x.value for x in v)))

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.gj_destroy(self._ctx)
            self._ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        name = C.create_string_buffer(256)
        cus, mem = C.c_int(0), C.c_uint64(0)
        self._check(self._lib.gj_device_info(self._ctx, name, 256, C.byref(cus), C.byref(mem)))
        return {"name": name.value.decode(), "compute_units": cus.value, "hbm_bytes": mem.value}

    def identity(self) -> str:
        """Which physical GPU this context runs on: ``pci=<bus id> uuid=<hex> hip=<index>`` (gj_device_identity)."""
        buf = C.create_string_buffer(160)
        self._check(self._lib.gj_device_identity(self._ctx, buf, 160))
        return buf.value.decode()

    def set_stream(self, stream_handle: Optional[int], external: bool = True):
        """Run on an external HIP stream (``torch.cuda.current_stream().cuda_stream``; 0 is
        the legacy default stream).  ``external=False`` returns to the context's own stream."""
        self._check(self._lib.gj_set_stream(self._ctx, stream_handle or None, 1 if external else 0))

    def synchronize(self):
        self._check(self._lib.gj_synchronize(self._ctx))

    def set_unpack(self, offset: float = 127.5, scale: float = 1.0 / 127.5):
        """sample = (u8 - offset) * scale for every later call on this context (gj_set_unpack):
        (127.5, 1/127.5) is the Python reference's convention, (128, 1/128) gnssdec's."""
        self._check(self._lib.gj_set_unpack(self._ctx, float(offset), float(scale)))

    def set_fill_threads(self, n: int = 0):
        """Host threads per staged copy (gj_set_fill_threads): 0 = by capture size, 1..16 = that many."""
        self._check(self._lib.gj_set_fill_threads(self._ctx, int(n)))

    def get_unpack(self):
        o, s = C.c_double(0), C.c_double(0)
        self._check(self._lib.gj_get_unpack(self._ctx, C.byref(o), C.byref(s)))
        return o.value, s.value

    def reserve(self, nbytes: int):
        self._check(self._lib.gj_reserve(self._ctx, int(nbytes)))

    def alloc(self, nbytes: int) -> DevBuf:
        return DevBuf(self, nbytes)

    def timer_start(self):
        self._check(self._lib.gj_timer_start(self._ctx))

    def timer_stop(self) -> float:
        ms = C.c_float(0)
        self._check(self._lib.gj_timer_stop(self._ctx, C.byref(ms)))
        return ms.value

    # ------------------------------------------------------------------ host arrays
    def chunk_power(self, raw, chunk_bytes: int = 65536, eps: float = 1e-10,
                    odd_chunk_zero: bool = False) -> np.ndarray:
        hit = self._cached(raw, ("chunk_power", int(chunk_bytes), float(np.float32(eps)), bool(odd_chunk_zero)))
        if hit is not None:
            return hit                                       # read-only array computed while the capture uploaded
        self._count("chunk_power")
        ptr, nbytes, _keep = self._input(raw)
        n = self._lib.gj_chunk_count(nbytes, chunk_bytes)
        out = np.empty(n, np.float32)
        n_out, ms = C.c_size_t(0), C.c_float(0)
        self._check(self._lib.gj_chunk_power_u8(
            self._ctx, ptr, nbytes, chunk_bytes, eps,
            _ffi.GJ_CP_ODD_CHUNK_ZERO if odd_chunk_zero else 0, out.ctypes.data, out.size,
            C.byref(n_out), C.byref(ms)))
        self.last_kernel_ms = ms.value
        return out[:n_out.value]

    def welch(self, raw, chunk_samples: int = 2048000, nperseg: int = 1024, fs: float = 2.048e6,
              shift: bool = True, want_db: bool = True):
        """(psd[rows, nperseg], psd_db[rows, nperseg] | None), float32."""
        key = ("welch", int(chunk_samples), int(nperseg), float(fs), bool(shift))
        if isinstance(raw, Capture) and raw.results.get(key) is not None and (raw.results[key][1] is not None or not want_db):
            hit = self._cached(raw, key)
            if hit is not None:
                return hit[0], (hit[1] if want_db else None)   # read-only arrays computed while the capture uploaded
        self._count("welch")
        ptr, nbytes, _keep = self._input(raw)
        rows = self._lib.gj_welch_rows(nbytes, chunk_samples, nperseg)
        psd = np.empty((rows, nperseg), np.float32)
        db = np.empty((rows, nperseg), np.float32) if want_db else None
        rows_out, ms = C.c_size_t(0), C.c_float(0)
        self._check(self._lib.gj_welch_u8(
            self._ctx, ptr, nbytes, chunk_samples, nperseg, fs,
            _ffi.GJ_WELCH_SHIFT if shift else 0, psd.ctypes.data,
            db.ctypes.data if want_db else None, psd.size, C.byref(rows_out), C.byref(ms)))
        self.last_kernel_ms = ms.value
        return psd, db

    def amp_stats(self, raw, threshold: float) -> AmpStats:
        hit = self._cached(raw, ("amp_stats", float(np.float32(threshold))))
        if hit is not None:
            return AmpStats.from_buffer_copy(bytes(hit))
        self._count("amp_stats")
        ptr, nbytes, _keep = self._input(raw)
        out, ms = AmpStats(), C.c_float(0)
        self._check(self._lib.gj_amp_stats_u8(self._ctx, ptr, nbytes, threshold,
                                              C.byref(out), C.byref(ms)))
        self.last_kernel_ms = ms.value
        return out

    def onset(self, raw, noise_samples: int = 200000, window: int = 1000,
              factor: float = 50.0) -> Onset:
        hit = self._cached(raw, ("onset", int(noise_samples), int(window), float(np.float32(factor))))
        if hit is not None:
            return Onset.from_buffer_copy(bytes(hit))
        self._count("onset")
        ptr, nbytes, _keep = self._input(raw)
        out, ms = Onset(), C.c_float(0)
        self._check(self._lib.gj_onset_u8(self._ctx, ptr, nbytes, noise_samples,
                                          window, factor, C.byref(out), C.byref(ms)))
        self.last_kernel_ms = ms.value
        return out

    def xcorr_lags_at(self, captures: Sequence["Capture"], starts: Sequence[int], n_samples: int,
                      pairs: Sequence[Sequence[int]], want_margins: bool = False):
        """Lags between resident captures: slice a = n_samples I/Q pairs of captures[a] from
        starts[a] (negative / out of range -> GJ_LAG_INVALID for its pairs).  No slice ever
        leaves HBM: the slices go to the library as device addresses inside the captures."""
        prs = [tuple(int(x) for x in p) for p in np.asarray(pairs, np.int64).reshape(-1, 2)]
        ok = [0 <= int(st) and 2 * (int(st) + n_samples) <= c.nbytes for c, st in zip(captures, starts)]
        lags = np.full(len(prs), GJ_LAG_INVALID, np.int32)
        peaks = np.zeros(len(prs), np.float32)
        margins = np.zeros(len(prs), np.float32)
        live = [k for k, (i, j) in enumerate(prs) if ok[i] and ok[j]]
        if live and n_samples > 0:
            used = sorted({a for k in live for a in prs[k]})
            local = {a: n for n, a in enumerate(used)}
            ptrs = (C.c_void_p * len(used))(*[captures[a].ptr + 2 * int(starts[a]) for a in used])
            flat = np.ascontiguousarray(np.array([[local[prs[k][0]], local[prs[k][1]]] for k in live], np.int32).reshape(-1))
            l2, p2, m2 = (np.empty(len(live), np.int32), np.empty(len(live), np.float32), np.empty(len(live), np.float32))
            ms = C.c_float(0)
            self._check(self._lib.gj_xcorr_lags_u8(
                self._ctx, ptrs, len(used), n_samples, flat.ctypes.data_as(C.POINTER(C.c_int32)), len(live),
                l2.ctypes.data_as(C.POINTER(C.c_int32)), p2.ctypes.data_as(C.POINTER(C.c_float)),
                m2.ctypes.data_as(C.POINTER(C.c_float)), C.byref(ms)))
            self.last_kernel_ms = ms.value
            lags[live], peaks[live], margins[live] = l2, p2, m2
        if want_margins:
            return lags, peaks, margins
        return lags, peaks

    def byte_histogram(self, cap: "Capture", chunk_samples: int = 2048000, nperseg: int = 1024,
                       stride: int = 100) -> np.ndarray:
        """256-bin histogram of raw_chunk[::stride] over the chunks the waterfall keeps
        (widmo_plot.py:35,85)."""
        d = DevBuf(self, 8 * 256)          # a buffer of this call's own: several threads may histogram at once
        try:
            self.byte_histogram_dev(cap, cap.nbytes, chunk_samples, nperseg, stride, d)
            return d.download(np.uint64, 256)
        finally:
            d.free()

    def xcorr_lags(self, slices: Sequence, pairs: Sequence[Sequence[int]], want_margins: bool = False):
        """lags[p], peaks[p] (, margins[p]) for pairs (i, j): lag of slice j relative to slice i
        (= argmax|correlate(slice_j, slice_i, 'full')| - (N-1))."""
        arrs = [as_u8(s) for s in slices]
        n = arrs[0].size // 2
        if any(a.size != 2 * n for a in arrs):
            raise ValueError("slices must have equal length")
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        flat = np.ascontiguousarray(np.asarray(pairs, np.int32).reshape(-1))
        npairs = flat.size // 2
        lags = np.empty(npairs, np.int32)
        peaks = np.empty(npairs, np.float32)
        margins = np.empty(npairs, np.float32)
        ms = C.c_float(0)
        self._check(self._lib.gj_xcorr_lags_u8(
            self._ctx, ptrs, len(arrs), n, flat.ctypes.data_as(C.POINTER(C.c_int32)), npairs,
            lags.ctypes.data_as(C.POINTER(C.c_int32)), peaks.ctypes.data_as(C.POINTER(C.c_float)),
            margins.ctypes.data_as(C.POINTER(C.c_float)), C.byref(ms)))
        self.last_kernel_ms = ms.value
        if want_margins:
            return lags, peaks, margins
        return lags, peaks

    # ------------------------------------------------------------------ device pointers
    def chunk_count(self, nbytes: int, chunk_bytes: int) -> int:
        return self._lib.gj_chunk_count(nbytes, chunk_bytes)

    def welch_rows(self, nbytes: int, chunk_samples: int, nperseg: int) -> int:
        return self._lib.gj_welch_rows(nbytes, chunk_samples, nperseg)

    def welch_workspace(self, nbytes: int, chunk_samples: int, nperseg: int) -> int:
        return self._lib.gj_welch_workspace(self._ctx, nbytes, chunk_samples, nperseg)

    def xcorr_workspace(self, n_ant: int, n_samples: int, n_pairs: int) -> int:
        return self._lib.gj_xcorr_workspace(self._ctx, n_ant, n_samples, n_pairs)

    def chunk_power_dev(self, d_iq, nbytes, chunk_bytes, d_power, eps=1e-10, flags=0):
        self._check(self._lib.gj_chunk_power_dev(self._ctx, _ptr(d_iq), nbytes, chunk_bytes, eps,
                                                 flags, _ptr(d_power)))

    def power_threshold_dev(self, d_power, n, d_stats, d_mask=None, pct=5.0, rise_db=6.0):
        self._check(self._lib.gj_power_threshold_dev(self._ctx, _ptr(d_power), n, pct, rise_db,
                                                     _ptr(d_stats), _ptr(d_mask) or None))

    def welch_dev(self, d_iq, nbytes, chunk_samples, nperseg, fs, d_psd, d_psd_db=None, shift=True):
        self._check(self._lib.gj_welch_dev(self._ctx, _ptr(d_iq), nbytes, chunk_samples, nperseg,
                                           fs, _ffi.GJ_WELCH_SHIFT if shift else 0, _ptr(d_psd),
                                           _ptr(d_psd_db) or None))

    def welch_batch_dev(self, d_iqs, nbytes_each, chunk_samples, nperseg, fs, d_psds, shift=True):
        """K2 of several captures of ONE length in one launch + one finalize (gj_welch_batch_dev); same bits per capture
        as welch_dev."""
        n = len(d_iqs)
        iq = (C.c_void_p * n)(*[_ptr(p) for p in d_iqs])
        out = (C.c_void_p * n)(*[_ptr(p) for p in d_psds])
        self._check(self._lib.gj_welch_batch_dev(self._ctx, iq, n, int(nbytes_each), chunk_samples, nperseg, fs,
                                                 _ffi.GJ_WELCH_SHIFT if shift else 0, out))

    def pack_results_dev(self, captures, nperseg, d_pairs=None, d_lags=None, d_peaks=None, d_margins=None):
        """One result vector per capture in ONE launch (gj_pack_results_dev); ``captures``: list of _ffi.CombineCapture."""
        ca = (_ffi.CombineCapture * len(captures))(*captures)
        self._check(self._lib.gj_pack_results_dev(self._ctx, ca, len(captures), nperseg, _ptr(d_pairs) or None, _ptr(d_lags) or None,
                                                  _ptr(d_peaks) or None, _ptr(d_margins) or None))

    def welch_timed_dev(self, d_iq, nbytes, chunk_samples, nperseg, fs, d_psd, d_psd_db=None, shift=True):
        """welch_dev with events around the transform kernel and around the finalize launch; synchronises.
        Returns (kernel_ms, finalize_ms) (gj_welch_timed_dev)."""
        k, f = C.c_float(0), C.c_float(0)
        self._check(self._lib.gj_welch_timed_dev(self._ctx, _ptr(d_iq), nbytes, chunk_samples, nperseg, fs,
                                                 _ffi.GJ_WELCH_SHIFT if shift else 0, _ptr(d_psd), _ptr(d_psd_db) or None,
                                                 C.byref(k), C.byref(f)))
        return float(k.value), float(f.value)

    def byte_histogram_dev(self, d_iq, nbytes, chunk_samples, nperseg, stride, d_hist):
        self._check(self._lib.gj_byte_histogram_dev(self._ctx, _ptr(d_iq), nbytes, chunk_samples,
                                                    nperseg, stride, _ptr(d_hist)))

    def amp_stats_dev(self, d_iq, nbytes, threshold, d_out):
        self._check(self._lib.gj_amp_stats_dev(self._ctx, _ptr(d_iq), nbytes, threshold, _ptr(d_out)))

    def onset_dev(self, d_iq, nbytes, noise_samples, window, factor, d_out):
        self._check(self._lib.gj_onset_dev(self._ctx, _ptr(d_iq), nbytes, noise_samples, window,
                                           factor, _ptr(d_out)))

    def stream_scan_dev(self, d_iq, nbytes, chunk_bytes, d_power, rssi_threshold, d_amp, noise_samples, window,
                        factor, d_onset, eps=1e-10, flags=0):
        """K1 + K3 + K4 in one pass over the capture (gj_stream_scan_dev)."""
        self._check(self._lib.gj_stream_scan_dev(self._ctx, _ptr(d_iq), nbytes, chunk_bytes, eps, flags,
                                                 _ptr(d_power), rssi_threshold, _ptr(d_amp), noise_samples,
                                                 window, factor, _ptr(d_onset)))

    @staticmethod
    def _scan_extra(d_stats, d_mask, pct, rise_db, slice_samples, d_slot):
        return _ffi.ScanExtra(float(pct), float(rise_db), _ptr(d_stats) or None, _ptr(d_mask) or None,
                              int(slice_samples) if d_slot is not None else 0, _ptr(d_slot) or None)

    def capture_scan_dev(self, d_iq, nbytes, chunk_bytes, d_power, rssi_threshold, d_amp, noise_samples, window,
                         factor, d_onset, *, d_stats=None, d_mask=None, pct=5.0, rise_db=6.0, slice_samples=0, d_slot=None,
                         eps=1e-10, flags=0):
        """The per-capture side chain in two launches (gj_capture_scan_dev): the fused pass, then one tail launch with
        the noise-floor threshold (``d_stats`` / ``d_mask``), the amplitude totals, the onset record and the TDOA slot cut
        at that onset (``d_slot``, ``slice_samples``).  Same results as stream_scan_dev + power_threshold_dev +
        tdoa_slot_dev."""
        x = self._scan_extra(d_stats, d_mask, pct, rise_db, slice_samples, d_slot)
        self._check(self._lib.gj_capture_scan_dev(self._ctx, _ptr(d_iq), nbytes, chunk_bytes, eps, flags,
                                                  _ptr(d_power), rssi_threshold, _ptr(d_amp), noise_samples,
                                                  window, factor, _ptr(d_onset), C.byref(x)))

    def tdoa_slot_bytes(self, n_samples: int) -> int:
        return self._lib.gj_tdoa_slot_bytes(n_samples)

    def tdoa_slot_dev(self, d_iq, nbytes, d_start, n_samples, d_slot):
        """Cut the slice that starts at the DEVICE scalar *d_start into a TDOA slot."""
        self._check(self._lib.gj_tdoa_slot_dev(self._ctx, _ptr(d_iq), nbytes, _ptr(d_start), n_samples, _ptr(d_slot)))

    def xcorr_slots_dev(self, d_slots, slot_stride, n_ant, n_samples, pairs, d_lags, d_peaks, d_margins=None):
        flat = np.ascontiguousarray(np.asarray(pairs, np.int32).reshape(-1))
        self._check(self._lib.gj_xcorr_slots_dev(
            self._ctx, _ptr(d_slots), slot_stride, n_ant, n_samples,
            flat.ctypes.data_as(C.POINTER(C.c_int32)), flat.size // 2, _ptr(d_lags), _ptr(d_peaks),
            _ptr(d_margins) or None))

    def xcorr_lags_dev(self, d_iqs, nbytes_list, d_starts, n_samples, pairs, d_lags, d_peaks, d_margins=None):
        n_ant = len(d_iqs)
        ptrs = (C.c_void_p * n_ant)(*[_ptr(p) for p in d_iqs])
        sizes = (C.c_size_t * n_ant)(*[int(b) for b in nbytes_list])
        flat = np.ascontiguousarray(np.asarray(pairs, np.int32).reshape(-1))
        self._check(self._lib.gj_xcorr_lags_dev(
            self._ctx, ptrs, sizes, n_ant, _ptr(d_starts), n_samples,
            flat.ctypes.data_as(C.POINTER(C.c_int32)), flat.size // 2, _ptr(d_lags), _ptr(d_peaks),
            _ptr(d_margins) or None))

    def pack_result_dev(self, n_chunks, d_power, d_stats, d_amp, d_onset, d_psd, rows, nperseg, rank, n_pairs,
                        pair_capacity, d_pairs, d_lags, d_peaks, d_margins, d_out):
        self._check(self._lib.gj_pack_result_dev(self._ctx, n_chunks, _ptr(d_power), _ptr(d_stats), _ptr(d_amp),
                                                 _ptr(d_onset), _ptr(d_psd), rows, nperseg, rank, n_pairs, pair_capacity,
                                                 _ptr(d_pairs) or None, _ptr(d_lags) or None, _ptr(d_peaks) or None,
                                                 _ptr(d_margins) or None, _ptr(d_out)))

    # ------------------------------------------------------------------ one capture over several GPUs
    def amp_tile_count(self, nbytes: int) -> int:
        return self._lib.gj_amp_tile_count(int(nbytes))

    def part_scan_dev(self, view, chunk_bytes, d_power, rssi_threshold, d_tiles, d_amp, noise_samples, window, factor,
                      d_onset, eps=1e-10, flags=0):
        """K1 + K3 + K4 of one part (gj_part_scan_dev); ``view`` is a _ffi.PartView."""
        self._check(self._lib.gj_part_scan_dev(self._ctx, C.byref(view), chunk_bytes, eps, flags, _ptr(d_power),
                                               rssi_threshold, _ptr(d_tiles), _ptr(d_amp), noise_samples, window, factor,
                                               _ptr(d_onset)))

    def part_capture_scan_dev(self, view, chunk_bytes, d_power, rssi_threshold, d_tiles, d_amp, noise_samples, window,
                              factor, d_onset, *, slice_samples=0, d_slot=None, eps=1e-10, flags=0):
        """part_scan_dev + part_slot_dev (at the part's own onset) in two launches (gj_part_capture_scan_dev)."""
        x = self._scan_extra(None, None, 5.0, 6.0, slice_samples, d_slot)
        self._check(self._lib.gj_part_capture_scan_dev(self._ctx, C.byref(view), chunk_bytes, eps, flags, _ptr(d_power),
                                                       rssi_threshold, _ptr(d_tiles), _ptr(d_amp), noise_samples, window,
                                                       factor, _ptr(d_onset), C.byref(x)))

    def part_welch_dev(self, view, chunk_samples, nperseg, fs, d_psd, d_psd_db=None, shift=True):
        self._check(self._lib.gj_part_welch_dev(self._ctx, C.byref(view), chunk_samples, nperseg, fs,
                                                _ffi.GJ_WELCH_SHIFT if shift else 0, _ptr(d_psd), _ptr(d_psd_db) or None))

    def part_welch_workspace(self, view, chunk_samples, nperseg) -> int:
        return self._lib.gj_part_welch_workspace(self._ctx, C.byref(view), chunk_samples, nperseg)

    def part_slot_dev(self, view, d_start, n_samples, d_slot):
        self._check(self._lib.gj_part_slot_dev(self._ctx, C.byref(view), _ptr(d_start), n_samples, _ptr(d_slot)))

    def slots_pick_dev(self, d_slots, slot_stride, d_offsets, d_members, n_groups, d_out):
        self._check(self._lib.gj_slots_pick_dev(self._ctx, _ptr(d_slots), slot_stride, _ptr(d_offsets), _ptr(d_members),
                                                n_groups, _ptr(d_out)))

    def amp_combine_dev(self, d_tiles, n_tiles, d_parts, n_parts, total_bytes, d_out):
        self._check(self._lib.gj_amp_combine_dev(self._ctx, _ptr(d_tiles), n_tiles, _ptr(d_parts), n_parts, total_bytes,
                                                 _ptr(d_out)))

    def onset_combine_dev(self, d_parts, n_parts, d_out):
        self._check(self._lib.gj_onset_combine_dev(self._ctx, _ptr(d_parts), n_parts, _ptr(d_out)))

    def part_result_len(self, chunk_cap, tile_cap, rows_cap, nperseg, pair_cap) -> int:
        return self._lib.gj_part_result_len(chunk_cap, tile_cap, rows_cap, nperseg, pair_cap)

    def pack_part_dev(self, args, d_out):
        self._check(self._lib.gj_pack_part_dev(self._ctx, C.byref(args), _ptr(d_out)))

    def combine_plan(self, copies, captures, rows_bytes, arena, nperseg, d_pairs, d_lags, d_peaks, d_margins,
                     pct=5.0, rise_db=6.0):
        """gj_combine_plan_create: ``copies`` / ``captures`` are lists of _ffi.CombineCopy / _ffi.CombineCapture; ``arena``
        the one device allocation (torch uint8 tensor or DevBuf) every destination lies in.  Returns the plan handle."""
        cs = (_ffi.CombineCopy * len(copies))(*copies)
        ca = (_ffi.CombineCapture * len(captures))(*captures)
        nbytes = arena.numel() if hasattr(arena, "numel") else arena.nbytes
        h = C.c_void_p()
        self._check(self._lib.gj_combine_plan_create(self._ctx, cs, len(copies), ca, len(captures), int(rows_bytes), _ptr(arena),
                                                     int(nbytes), int(nperseg), pct, rise_db, _ptr(d_pairs), _ptr(d_lags),
                                                     _ptr(d_peaks), _ptr(d_margins), C.byref(h)))
        return h

    def split_combine_dev(self, plan, d_rows):
        """Every capture of a split run rebuilt and finished in three launches (gj_split_combine_dev)."""
        self._check(self._lib.gj_split_combine_dev(self._ctx, plan, _ptr(d_rows)))

    def combine_plan_destroy(self, plan):
        if getattr(self, "_ctx", None):
            self._lib.gj_combine_plan_destroy(self._ctx, plan)

    def synth_dev(self, spec, n_samples: int, d_out, first_sample: int = 0):
        """Fill d_out[2*n_samples] with the capture described by a synth.StreamSpec."""
        p = SynthParams(spec.key_noise, spec.key_common, spec.delay, spec.jam_start,
                        min(spec.jam_end, (1 << 62)), spec.noise_k, spec.jam_k, spec.dc_i_q8,
                        spec.dc_q_q8)
        self._check(self._lib.gj_synth_u8_dev(self._ctx, C.byref(p), first_sample, n_samples,
                                              _ptr(d_out)))
