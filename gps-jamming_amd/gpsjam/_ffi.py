"""ctypes binding of libgpsjam_hip.so (include/gpsjam.h).

There is no CPU fallback: if the shared library is missing or cannot be loaded the
import of anything that needs it raises ``GpsJamLibraryError`` with the build hint.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GPSJAM_LIB lets a developer A/B an alternative build of the same library (tools/ab_build.sh)
LIB_PATH = os.environ.get("GPSJAM_LIB") or os.path.join(os.path.dirname(_HERE), "csrc", "libgpsjam_hip.so")


class GpsJamLibraryError(RuntimeError):
    pass


class GpsJamError(RuntimeError):
    def __init__(self, status: int, text: str):
        super().__init__(f"gpsjam: {text} (status {status})")
        self.status = status


class AmpStats(C.Structure):
    _fields_ = [("first_index", C.c_int64), ("count", C.c_uint64), ("sum", C.c_double),
                ("mean", C.c_float), ("reserved", C.c_float)]


class Onset(C.Structure):
    _fields_ = [("start_index", C.c_int64), ("noise_power", C.c_float),
                ("threshold", C.c_float), ("margin_hit", C.c_float),
                ("margin_before", C.c_float), ("guard_index", C.c_int64)]

    NEAR_TIE = 1e-6       # the rounding band of include/gpsjam.h (gj_onset.guard_index)

    @property
    def margin(self) -> float:
        """Smaller of the two decision margins (see include/gpsjam.h, gj_onset)."""
        if self.start_index < 0:
            return float(self.margin_before)
        return float(min(self.margin_hit, self.margin_before))

    @property
    def near_tie(self) -> bool:
        """True when the reference's float32 arithmetic could decide differently: a moving average sits
        inside the rounding band in front of the crossing, or the crossing itself clears the threshold by
        less than the band."""
        if self.guard_index != self.start_index:
            return True
        return self.start_index >= 0 and self.margin_hit < self.NEAR_TIE


class IngestPlan(C.Structure):
    """gj_ingest_plan (include/gpsjam.h): what gj_ingest_* computes while the capture is uploaded."""
    _fields_ = [("chunk_bytes", C.c_size_t), ("eps", C.c_float), ("power_flags", C.c_int),
                ("rssi_threshold", C.c_float), ("noise_samples", C.c_int), ("window", C.c_int), ("factor", C.c_float),
                ("chunk_samples", C.c_size_t), ("nperseg", C.c_int), ("welch_flags", C.c_int), ("fs", C.c_double)]


class IngestResult(C.Structure):
    _fields_ = [("nbytes", C.c_size_t), ("n_chunks", C.c_size_t), ("rows", C.c_size_t), ("amp", AmpStats),
                ("onset", Onset), ("upload_ms", C.c_float), ("total_ms", C.c_float)]


class ScanExtra(C.Structure):
    """gj_scan_extra: what the scan's tail launch does on top of K1 + K3 + K4 (threshold of the power map, TDOA slot)."""
    _fields_ = [("pct", C.c_float), ("rise_db", C.c_float), ("d_stats", C.c_void_p), ("d_mask", C.c_void_p),
                ("slice_samples", C.c_size_t), ("d_slot", C.c_void_p)]


class IngestJob(C.Structure):
    """gj_ingest_job: one file of gj_ingest_files."""
    _fields_ = [("path", C.c_char_p), ("offset", C.c_size_t), ("max_bytes", C.c_size_t), ("power", C.c_void_p),
                ("power_cap", C.c_size_t), ("psd", C.c_void_p), ("psd_db", C.c_void_p), ("psd_cap_floats", C.c_size_t),
                ("result", IngestResult), ("dptr", C.c_void_p), ("status", C.c_int)]


class PartView(C.Structure):
    """gj_part_view: one part of a capture split over GPUs (include/gpsjam.h)."""
    _fields_ = [("d_buf", C.c_void_p), ("buf_bytes", C.c_size_t), ("buf_first_byte", C.c_size_t),
                ("own_first_byte", C.c_size_t), ("own_bytes", C.c_size_t), ("total_bytes", C.c_size_t),
                ("d_noise", C.c_void_p)]


class AmpPart(C.Structure):
    _fields_ = [("first_index", C.c_int64), ("count", C.c_uint64), ("sum", C.c_double), ("tail", C.c_double)]


class PartPack(C.Structure):
    _fields_ = [("rank", C.c_int32), ("antenna", C.c_int32), ("part", C.c_int32), ("parts", C.c_int32),
                ("first_chunk", C.c_uint64), ("n_chunks", C.c_uint64), ("chunk_cap", C.c_uint64),
                ("first_row", C.c_uint64), ("rows", C.c_uint64), ("rows_cap", C.c_uint64),
                ("first_tile", C.c_uint64), ("n_tiles", C.c_uint64), ("tile_cap", C.c_uint64),
                ("first_sample", C.c_int64),
                ("nperseg", C.c_int32), ("n_pairs", C.c_int32), ("pair_cap", C.c_int32), ("reserved", C.c_int32),
                ("d_power", C.c_void_p), ("d_amp", C.c_void_p), ("d_onset", C.c_void_p), ("d_tiles", C.c_void_p),
                ("d_psd", C.c_void_p), ("d_pairs", C.c_void_p), ("d_lags", C.c_void_p), ("d_peaks", C.c_void_p),
                ("d_margins", C.c_void_p)]


class CombineCopy(C.Structure):
    """gj_combine_copy: one run of elements from the gathered part vectors into a capture-order array."""
    _fields_ = [("src_byte", C.c_uint64), ("dst", C.c_uint64), ("count", C.c_uint64), ("src_stride", C.c_uint32),
                ("kind", C.c_uint32)]


class CombineCapture(C.Structure):
    """gj_combine_capture: the arrays of one capture on the combining rank."""
    _fields_ = [("n_chunks", C.c_uint64), ("rows", C.c_uint64), ("n_tiles", C.c_uint64), ("total_bytes", C.c_uint64),
                ("n_parts", C.c_int32), ("antenna", C.c_int32), ("n_pairs", C.c_int32), ("pair_cap", C.c_int32),
                ("d_power", C.c_void_p), ("d_stats", C.c_void_p), ("d_tiles", C.c_void_p), ("d_amp_parts", C.c_void_p),
                ("d_onset_parts", C.c_void_p), ("d_amp", C.c_void_p), ("d_onset", C.c_void_p), ("d_psd", C.c_void_p),
                ("d_out", C.c_void_p)]


GJ_COPY_F64_F32, GJ_COPY_F64, GJ_COPY_F32, GJ_COPY_F64_I32 = 0, 1, 2, 3


class SynthParams(C.Structure):
    _fields_ = [("key_noise", C.c_uint64), ("key_common", C.c_uint64), ("delay", C.c_int64),
                ("jam_start", C.c_int64), ("jam_end", C.c_int64), ("noise_k", C.c_int32),
                ("jam_k", C.c_int32), ("dc_i_q8", C.c_int32), ("dc_q_q8", C.c_int32)]


GJ_CP_ODD_CHUNK_ZERO = 1
GJ_WELCH_SHIFT = 1
GJ_MAX_ANTENNAS = 16
GJ_LAG_INVALID = -(1 << 31)
GJ_SLOT_HEADER = 16
GJ_COMM_ID_BYTES = 128
GJ_VERSION = 150

_vp, _sz, _i, _f, _d = C.c_void_p, C.c_size_t, C.c_int, C.c_float, C.c_double
_pf, _psz = C.POINTER(C.c_float), C.POINTER(C.c_size_t)

# name -> (restype, argtypes); every symbol include/gpsjam.h declares
SIGNATURES = {
    "gj_version": (_i, []),
    "gj_strerror": (C.c_char_p, [_i]),
    "gj_last_error": (C.c_char_p, [_vp]),
    "gj_device_count": (_i, [C.POINTER(_i)]),
    "gj_create": (_i, [_i, C.POINTER(_vp)]),
    "gj_destroy": (_i, [_vp]),
    "gj_set_stream": (_i, [_vp, _vp, _i]),
    "gj_synchronize": (_i, [_vp]),
    "gj_set_unpack": (_i, [_vp, _d, _d]),
    "gj_set_fill_threads": (_i, [_vp, _i]),
    "gj_get_unpack": (_i, [_vp, C.POINTER(_d), C.POINTER(_d)]),
    "gj_device_info": (_i, [_vp, C.c_char_p, _sz, C.POINTER(_i), C.POINTER(C.c_uint64)]),
    "gj_device_identity": (_i, [_vp, C.c_char_p, _sz]),
    "gj_reserve": (_i, [_vp, _sz]),
    "gj_debug_set_wait_hook": (_i, [_vp, _vp, _vp]),
    "gj_probe_busy_dev": (_i, [_vp, _f]),
    "gj_debug_inject": (_i, [_vp, _i, _i]),
    "gj_debug_counters": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "gj_malloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "gj_free": (_i, [_vp, _vp]),
    "gj_memcpy_h2d": (_i, [_vp, _vp, _vp, _sz]),
    "gj_memcpy_d2h": (_i, [_vp, _vp, _vp, _sz]),
    "gj_upload": (_i, [_vp, _vp, _sz, C.POINTER(_vp)]),
    "gj_upload_file": (_i, [_vp, C.c_char_p, _sz, _sz, C.POINTER(_vp), _psz]),
    "gj_ingest_u8": (_i, [_vp, _vp, _sz, C.POINTER(IngestPlan), _vp, _sz, _vp, _vp, _sz, C.POINTER(IngestResult),
                          C.POINTER(_vp)]),
    "gj_ingest_files": (_i, [_vp, C.POINTER(IngestJob), _i, C.POINTER(IngestPlan)]),
    "gj_ingest_file": (_i, [_vp, C.c_char_p, _sz, _sz, C.POINTER(IngestPlan), _vp, _sz, _vp, _vp, _sz,
                            C.POINTER(IngestResult), C.POINTER(_vp)]),
    "gj_timer_start": (_i, [_vp]),
    "gj_timer_stop": (_i, [_vp, _pf]),
    "gj_chunk_count": (_sz, [_sz, _sz]),
    "gj_chunk_power_dev": (_i, [_vp, _vp, _sz, _sz, _f, _i, _vp]),
    "gj_chunk_power_u8": (_i, [_vp, _vp, _sz, _sz, _f, _i, _vp, _sz, _psz, _pf]),
    "gj_power_threshold_dev": (_i, [_vp, _vp, _sz, _f, _f, _vp, _vp]),
    "gj_welch_rows": (_sz, [_sz, _sz, _i]),
    "gj_welch_dev": (_i, [_vp, _vp, _sz, _sz, _i, _d, _i, _vp, _vp]),
    "gj_welch_batch_dev": (_i, [_vp, C.POINTER(_vp), _i, _sz, _sz, _i, _d, _i, C.POINTER(_vp)]),
    "gj_welch_timed_dev": (_i, [_vp, _vp, _sz, _sz, _i, _d, _i, _vp, _vp, C.POINTER(_f), C.POINTER(_f)]),
    "gj_welch_u8": (_i, [_vp, _vp, _sz, _sz, _i, _d, _i, _vp, _vp, _sz, _psz, _pf]),
    "gj_welch_workspace": (_sz, [_vp, _sz, _sz, _i]),
    "gj_byte_histogram_dev": (_i, [_vp, _vp, _sz, _sz, _i, _i, _vp]),
    "gj_amp_stats_dev": (_i, [_vp, _vp, _sz, _f, _vp]),
    "gj_amp_stats_u8": (_i, [_vp, _vp, _sz, _f, C.POINTER(AmpStats), _pf]),
    "gj_onset_dev": (_i, [_vp, _vp, _sz, _i, _i, _f, _vp]),
    "gj_onset_u8": (_i, [_vp, _vp, _sz, _i, _i, _f, C.POINTER(Onset), _pf]),
    "gj_stream_scan_dev": (_i, [_vp, _vp, _sz, _sz, _f, _i, _vp, _f, _vp, _i, _i, _f, _vp]),
    "gj_capture_scan_dev": (_i, [_vp, _vp, _sz, _sz, _f, _i, _vp, _f, _vp, _i, _i, _f, _vp, C.POINTER(ScanExtra)]),
    "gj_xcorr_lags_dev": (_i, [_vp, C.POINTER(_vp), _psz, _i, _vp, _sz, C.POINTER(C.c_int32), _i,
                               _vp, _vp, _vp]),
    "gj_xcorr_lags_u8": (_i, [_vp, C.POINTER(_vp), _i, _sz, C.POINTER(C.c_int32), _i,
                              C.POINTER(C.c_int32), _pf, _pf, _pf]),
    "gj_tdoa_slot_bytes": (_sz, [_sz]),
    "gj_tdoa_slot_dev": (_i, [_vp, _vp, _sz, _vp, _sz, _vp]),
    "gj_xcorr_slots_dev": (_i, [_vp, _vp, _sz, _i, _sz, C.POINTER(C.c_int32), _i, _vp, _vp, _vp]),
    "gj_amp_tile_count": (_sz, [_sz]),
    "gj_part_scan_dev": (_i, [_vp, C.POINTER(PartView), _sz, _f, _i, _vp, _f, _vp, _vp, _i, _i, _f, _vp]),
    "gj_part_capture_scan_dev": (_i, [_vp, C.POINTER(PartView), _sz, _f, _i, _vp, _f, _vp, _vp, _i, _i, _f, _vp,
                                      C.POINTER(ScanExtra)]),
    "gj_part_welch_dev": (_i, [_vp, C.POINTER(PartView), _sz, _i, _d, _i, _vp, _vp]),
    "gj_part_welch_workspace": (_sz, [_vp, C.POINTER(PartView), _sz, _i]),
    "gj_part_slot_dev": (_i, [_vp, C.POINTER(PartView), _vp, _sz, _vp]),
    "gj_slots_pick_dev": (_i, [_vp, _vp, _sz, _vp, _vp, _i, _vp]),
    "gj_amp_combine_dev": (_i, [_vp, _vp, _sz, _vp, _i, _sz, _vp]),
    "gj_onset_combine_dev": (_i, [_vp, _vp, _i, _vp]),
    "gj_part_result_len": (_sz, [_sz, _sz, _sz, _i, _i]),
    "gj_pack_part_dev": (_i, [_vp, C.POINTER(PartPack), _vp]),
    "gj_combine_plan_create": (_i, [_vp, C.POINTER(CombineCopy), _i, C.POINTER(CombineCapture), _i, _sz, _vp, _sz, _i, _f, _f,
                                    _vp, _vp, _vp, _vp, C.POINTER(_vp)]),
    "gj_split_combine_dev": (_i, [_vp, _vp, _vp]),
    "gj_combine_plan_destroy": (_i, [_vp, _vp]),
    "gj_pack_results_dev": (_i, [_vp, C.POINTER(CombineCapture), _i, _i, _vp, _vp, _vp, _vp]),
    "gj_combine_plan_check": (_i, [C.POINTER(CombineCopy), _i, C.POINTER(CombineCapture), _i, _sz, _vp, _sz, _i, _i]),
    "gj_acq_search_dev": (_i, [_vp, _vp, _sz, _sz, _i, _i, _vp, _i, _vp, _i, _i, _d, _f, _vp, _vp]),
    "gj_acq_workspace": (_sz, [_vp, _i, _i, _i, _i, _i]),
    "gj_comm_unique_id": (_i, [_vp]),
    "gj_comm_init_rank": (_i, [_vp, _vp, _i, _i, C.POINTER(_vp)]),
    "gj_comm_rank": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "gj_comm_device": (_i, [_vp, C.POINTER(_i)]),
    "gj_comm_gather_dev": (_i, [_vp, _vp, _sz, _vp, _i]),
    "gj_comm_allgather_dev": (_i, [_vp, _vp, _sz, _vp]),
    "gj_comm_bcast_dev": (_i, [_vp, _vp, _sz, _i]),
    "gj_comm_destroy": (_i, [_vp]),
    "gj_xcorr_workspace": (_sz, [_vp, _i, _sz, _i]),
    "gj_pack_result_dev": (_i, [_vp, _sz, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "gj_synth_u8_dev": (_i, [_vp, C.POINTER(SynthParams), C.c_int64, _sz, _vp]),
}

_lib = None


def _share_torch_hip_runtime():
    """One HIP runtime per process.  A PyTorch-ROCm wheel bundles its own libamdhip64.so / libhsa-runtime64.so (same
    sonames as /opt/rocm's, which libgpsjam_hip.so is linked against).  If torch is imported FIRST the loader gives
    this library torch's copy (soname match) and all is well; the other way round torch loads a second runtime by file
    name and its device discovery fails ("no ROCm-capable device").  So when a torch with a bundled runtime is
    installed, that runtime is loaded here -- without importing torch -- before the library is, whatever the import
    order.  GPSJAM_SYSTEM_HIP=1 keeps /opt/rocm's (a process that never uses torch.cuda)."""
    import importlib.util
    import sys
    if os.environ.get("GPSJAM_SYSTEM_HIP") == "1" or "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if not os.path.exists(path):
            return
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:
            return


def load():
    """Load (once) and return the ctypes handle with typed signatures."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GpsJamLibraryError(
            f"{LIB_PATH} not found: build it with `make -C {os.path.dirname(LIB_PATH)}` "
            "(hipcc --offload-arch=gfx950) or `python -c 'import __graft_entry__ as g; g.build()'`. "
            "There is no CPU fallback.")
    _share_torch_hip_runtime()
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise GpsJamLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
