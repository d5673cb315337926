"""Host side of the batched GNSS acquisition search (``gj_acq_search_dev``).

The search itself -- carrier wipe-off, 2*nsamp-point FFT, product with the code spectrum, inverse
FFT, |.|^2 accumulation over up to ``intg`` milliseconds, peak test -- runs on the GPU for every PRN
and every Doppler bin at once.  What stays on the host is what the reference receiver computes once
at start-up, in a few microseconds per channel (GpsJammerApp/backend/):

* the spreading code of each PRN         sdrcode.c:102-149 (gencode_L1CA; IS-GPS-200 G1/G2 generators)
* its resampling to the sampling rate    sdrcmn.c:527-579  (rescode, the fixed-point SSE2 form) via sdrinit.c:439
* the Doppler bin list                   sdrinit.c:184,409-412 (+-7 kHz in 200 Hz steps: 71 bins)
* the mixer's phase index per sample     sdrcmn.c:618-659,676-684 (mixcarr, SSE2 form: 16-entry table,
                                         phases accumulated in doubles in ITS order, truncated toward zero)

Nothing here is a CPU fallback for the kernels: without the HIP library ``AcqSearch`` cannot be built.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Sequence

import numpy as np

CA_LEN = 1023                 # chips per C/A code period (LEN_L1CA)
CA_RATE = 1.023e6             # chips per second (CRATE_L1CA)
ACQ_INTG_L1CA = 10            # sdr.h:59
ACQ_HBAND = 7000.0            # sdr.h:64
ACQ_STEP = 200.0              # sdr.h:65
ACQ_THRESHOLD = 3.0           # sdr.h:66 (ACQTH)

# IS-GPS-200, Table 3-Ia: G2 code delay in chips for PRN 1..32 (the reference's table, sdrcode.c:104-106,
# continues with the SBAS / QZSS entries, which the GPS search does not use)
G2_DELAY = (5, 6, 7, 8, 17, 18, 139, 140, 141, 251, 252, 254, 255, 256, 257, 258, 469, 470, 471, 472,
            473, 474, 509, 512, 513, 514, 515, 516, 859, 860, 861, 862)


def _lfsr(taps: Sequence[int], n: int = CA_LEN) -> np.ndarray:
    """Output (stage 10) of a 10-stage shift register that starts as all ones; feedback = XOR of the
    stages in ``taps`` (1-based), as bits 0/1."""
    reg = 0x3FF                                   # bit k-1 = stage k
    out = np.empty(n, np.uint8)
    for i in range(n):
        out[i] = (reg >> 9) & 1
        fb = 0
        for t in taps:
            fb ^= (reg >> (t - 1)) & 1
        reg = ((reg << 1) | fb) & 0x3FF
    return out


_G1 = _lfsr((3, 10))
_G2 = _lfsr((2, 3, 6, 8, 9, 10))


def ca_code(prn: int) -> np.ndarray:
    """C/A code of GPS PRN 1..32 as int16 chips, +1 for a logical one (the reference's
    ``code[i] = -G1[i] * G2[i - delay]`` with its registers holding -1 for a one, sdrcode.c:129-145)."""
    if not 1 <= prn <= len(G2_DELAY):
        raise ValueError("GPS PRN must be 1..32")
    bits = _G1 ^ np.roll(_G2, G2_DELAY[prn - 1])
    return (2 * bits.astype(np.int16) - 1).astype(np.int16)


def resample_code(code: np.ndarray, nsamp: int, fs: float = 2.048e6, chip_rate: float = CA_RATE) -> np.ndarray:
    """rescode(code, len, coff=0, smax=0, ci=chip_rate/fs, n=nsamp) in the reference's 32-bit fixed-point form
    (sdrcmn.c:541-575): phase = round(k ci 2^nbit) for the first four samples, then += round(4 ci 2^nbit) per
    group of four, wrapped at len 2^nbit, index = phase >> nbit."""
    code = np.asarray(code, np.int16)
    ln = int(code.size)
    ci = (1.0 / fs) * chip_rate                               # sdr->ci = sdr->ti * sdr->crate (sdrinit.c:378,384)
    nbit = 31 - ln.bit_length() - 1                           # for (i = len, nbit = 31; i; i >>= 1, nbit--); nbit -= 1
    scale = 1 << nbit
    x = np.empty(4, np.int64)
    coff = 0.0
    for i in range(4):
        x[i] = int(coff * scale + 0.5)
        coff += ci
    step = int(ci * 4 * scale + 0.5)
    wrap = ln * scale
    out = np.empty(nsamp, np.int16)
    for g in range(0, nsamp, 4):
        x = np.where(x > wrap - 1, x - wrap, x)
        idx = x >> nbit
        out[g:g + 4] = code[idx[:min(4, nsamp - g)]]
        x = x + step
    return out


def doppler_bins(f_if: float = 0.0, hband: float = ACQ_HBAND, step: float = ACQ_STEP, foffset: float = 0.0) -> np.ndarray:
    """acq.freq[i] = f_if + (i - (nfreq-1)/2) step + foffset, nfreq = 2 (hband/step) + 1 (sdrinit.c:184,409-412;
    the reference's integer division of the two #defines is kept)."""
    nfreq = 2 * (int(hband) // int(step)) + 1
    i = np.arange(nfreq)
    return f_if + (i - (nfreq - 1) // 2) * step + foffset


def mixer_phase_table(freqs: Sequence[float], fs: float, m: int) -> np.ndarray:
    """uint8[n_freq][m]: the 4-bit table index mixcarr (SSE2 form, phi0 = 0) uses for sample n of a window of m
    samples at carrier ``freq``.  The reference keeps sixteen phases in doubles -- phi, phi + ps, then phi += 2 ps
    eight times -- adds 16 ps to each after every block of sixteen samples, and converts with cvttpd (truncation
    toward zero) before masking with 15 (sdrcmn.c:636-659,676-684,215-223); the same additions in the same order
    are made here so that every index is the reference's index, also where a phase sits on an integer."""
    if m % 16:
        raise ValueError("window length must be a multiple of 16")
    ti = 1.0 / fs
    out = np.empty((len(freqs), m), np.uint8)
    nblk = m // 16
    for f_idx, freq in enumerate(freqs):
        ps = float(freq) * 16 * ti
        base = np.empty(16, np.float64)
        phi = 0.0                                              # phi0 / DPI * 16 - floor(phi0 / DPI) * 16
        for k in range(0, 16, 2):
            base[k] = phi
            base[k + 1] = phi + ps
            phi += ps * 2
        inc = ps * 16
        ph = np.empty((nblk, 16), np.float64)
        ph[0] = base
        ph[1:] = inc
        ph = np.add.accumulate(ph, axis=0)                     # sequential double additions, block after block
        out[f_idx] = (np.trunc(ph).astype(np.int64) & 15).astype(np.uint8).reshape(-1)
    return out


@dataclass
class AcqResult:
    prn: int
    acquired: bool
    peak_ratio: float
    cn0: float
    code_index: int
    freq_index: int
    doppler_hz: float
    steps: int
    max_power: float
    second_power: float
    mean_power: float


class _AcqStruct(C.Structure):
    _fields_ = [("max_power", C.c_double), ("second_power", C.c_double), ("mean_power", C.c_double),
                ("peak_ratio", C.c_double), ("cn0", C.c_double), ("code_index", C.c_int32),
                ("freq_index", C.c_int32), ("steps", C.c_int32), ("acquired", C.c_int32)]


class AcqSearch:
    """Codes and mixer tables resident in HBM; ``search()`` runs one cold search of every PRN over every
    Doppler bin on a resident capture (what 32 channel threads of the reference do one after the other)."""

    def __init__(self, dev, prns: Sequence[int] = tuple(range(1, 33)), fs: float = 2.048e6, f_if: float = 0.0,
                 intg: int = ACQ_INTG_L1CA, threshold: float = ACQ_THRESHOLD, hband: float = ACQ_HBAND,
                 step: float = ACQ_STEP):
        self.dev, self.prns, self.fs, self.intg, self.threshold = dev, list(prns), fs, int(intg), float(threshold)
        self.ctime = CA_LEN / CA_RATE                          # sdr->ctime = clen / crate (sdrinit.c:385)
        self.nsamp = int(fs * self.ctime)                      # sdrinit.c:386
        self.nsampchip = int(self.nsamp / CA_LEN)              # sdrinit.c:387
        self.freqs = doppler_bins(f_if, hband, step)
        codes = np.stack([resample_code(ca_code(p), self.nsamp, fs) for p in self.prns]).astype(np.int16)
        phase = mixer_phase_table(self.freqs, fs, 2 * self.nsamp)
        self.d_codes = dev.alloc(codes.nbytes).upload(codes)
        self.d_phase = dev.alloc(phase.nbytes).upload(phase)
        self.d_out = dev.alloc(C.sizeof(_AcqStruct) * len(self.prns))
        dev.reserve(dev._lib.gj_acq_workspace(dev._ctx, self.nsamp, len(self.freqs), len(self.prns), self.intg, 0))

    def samples_needed(self) -> int:
        return (self.intg + 1) * self.nsamp

    def search_dev(self, d_iq, nbytes: int, first_sample: int = 0, d_power=None):
        """Enqueue one search (no host synchronisation); results land in ``self.d_out``."""
        from . import _ptr
        self.dev._check(self.dev._lib.gj_acq_search_dev(
            self.dev._ctx, _ptr(d_iq), int(nbytes), int(first_sample), self.nsamp, self.intg, self.d_codes.ptr,
            len(self.prns), self.d_phase.ptr, len(self.freqs), self.nsampchip, self.ctime, self.threshold,
            self.d_out.ptr, _ptr(d_power) or None))

    def results(self) -> List[AcqResult]:
        raw = self.d_out.download(np.uint8, C.sizeof(_AcqStruct) * len(self.prns)).tobytes()
        out = []
        for k, prn in enumerate(self.prns):
            r = _AcqStruct.from_buffer_copy(raw, k * C.sizeof(_AcqStruct))
            out.append(AcqResult(prn, bool(r.acquired), r.peak_ratio, r.cn0, r.code_index, r.freq_index,
                                 float(self.freqs[r.freq_index]), r.steps, r.max_power, r.second_power, r.mean_power))
        return out

    def search(self, capture, first_sample: int = 0, want_power: bool = False):
        """Search a resident capture (gpsjam.Capture / DevBuf / torch tensor with ``nbytes``)."""
        nbytes = getattr(capture, "nbytes", None)
        if nbytes is None:
            nbytes = capture.numel() * capture.element_size()
        d_power = None
        if want_power:
            d_power = self.dev.alloc(8 * len(self.prns) * len(self.freqs) * self.nsamp)
        self.dev.timer_start()
        self.search_dev(capture, nbytes, first_sample, d_power)
        self.dev.last_kernel_ms = self.dev.timer_stop()
        res = self.results()
        if want_power:
            p = d_power.download(np.float64).reshape(len(self.prns), len(self.freqs), self.nsamp)
            d_power.free()
            return res, p
        return res

    def close(self):
        for b in (self.d_codes, self.d_phase, self.d_out):
            b.free()
