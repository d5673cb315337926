"""CPU oracle for the jamming-detection DSP path  --  TEST INFRASTRUCTURE ONLY.

This module is a numpy / scipy.fft restatement of the reference's algorithm for the
hot path named in BASELINE.json.  It exists so that the HIP kernels can be checked
against something that is itself pinned to the reference.  It is NOT part of the
product: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  The product path (``gps-jamming_amd/``) never does and
fails loudly when the HIP library is missing.

Parity status: PINNED.  Every function below is compared in
``tests/golden/make_golden.py`` (run in the build container, where
``/root/reference`` is importable) against the reference's own functions on seeded
inputs, and the resulting vectors are committed under ``tests/golden/``;
``tests/test_oracle_golden.py`` re-checks the oracle against those vectors without
the reference.  The FFT/Welch/correlate arithmetic itself lives in third-party
scipy (un-vendored; reference pins ``scipy>=1.10.0``, ``numpy>=1.24.0`` in
requirements.txt:2-3; vectors were generated with scipy 1.15.3 / numpy 2.2.6).  The
published algorithms (Welch 1967 averaged periodogram as implemented by
``scipy.signal._spectral_py._spectral_helper``; FFT cross-correlation as implemented
by ``scipy.signal._signaltools.fftconvolve``) are restated here on top of
``scipy.fft`` (pocketfft) only.

All citations are ``path:line`` relative to the reference root.
"""
from __future__ import annotations

import math

import numpy as np
from scipy import fft as _sfft

# ----------------------------------------------------------------------------------
# constants of the reference (kept as defaults; every function takes them as arguments)
# ----------------------------------------------------------------------------------
POWER_CHUNK_SAMPLES = 32768          # GpsJammerApp/app/worker.py:82
POWER_RISE_DB = 6.0                  # worker.py:86
CIJ_CHUNK_BYTES = 131072             # GpsJammerApp/app/checkIfJamming.py:5
CIJ_CALIBRATION_FACTOR = 4.8         # checkIfJamming.py:95
SAMPLE_RATE = 2.048e6                # skrypty/widmo_plot.py:8, triangulateTDOA.py:13
WELCH_NPERSEG = 1024                 # widmo_plot.py:10
RSSI_TX_POWER = 40.0                 # skrypty/triangulateRSSI.py:9
RSSI_PATH_LOSS_EXP = 3.0             # triangulateRSSI.py:10
RSSI_FREQ_MHZ = 1575.42              # triangulateRSSI.py:11
RSSI_THRESHOLD = 0.1                 # triangulateRSSI.py:12
GRID_DENSITY = 300                   # triangulateRSSI.py:15
GRID_RANGE_MULT = 1.5                # triangulateRSSI.py:16
METERS_PER_DEG = 111320.0            # triangulateRSSI.py:19-20
TDOA_NOISE_SAMPLES = 200000          # skrypty/triangulateTDOA.py:21
TDOA_WINDOW = 1000                   # triangulateTDOA.py:22
TDOA_FACTOR = 50.0                   # triangulateTDOA.py:23
TDOA_SLICE = 50000                   # triangulateTDOA.py:26
SPEED_OF_LIGHT = 299792458           # triangulateTDOA.py:29


def _as_u8(raw) -> np.ndarray:
    a = np.asarray(raw)
    if a.dtype != np.uint8:
        raise TypeError("raw I/Q must be uint8")
    return a.reshape(-1)


# ----------------------------------------------------------------------------------
# (1) per-chunk power scan + threshold          worker.py:198-275
# ----------------------------------------------------------------------------------
def chunk_power(raw, chunk_bytes: int = 2 * POWER_CHUNK_SAMPLES) -> np.ndarray:
    """mean((I-127.5)^2 + (Q-127.5)^2) + 1e-10 for every ``chunk_bytes`` piece of the
    stream, ragged tail included (worker.py:216-230).  float32 in, float32 out: the
    mean of a float32 vector stays float32 (numpy pairwise summation)."""
    raw = _as_u8(raw)
    out = []
    for off in range(0, raw.size, chunk_bytes):
        piece = raw[off:off + chunk_bytes]
        centred = piece.astype(np.float32) - 127.5           # worker.py:222
        i_part = centred[0::2]
        q_part = centred[1::2]
        n = min(i_part.size, q_part.size)                    # worker.py:226
        p = i_part[:n] ** 2 + q_part[:n] ** 2                # worker.py:228
        with np.errstate(all="ignore"):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                out.append(np.mean(p) + 1e-10)               # worker.py:229
    return np.array(out)                                     # worker.py:239


def power_threshold(power_map: np.ndarray, rise_db: float = POWER_RISE_DB,
                    chunk_bytes: int = 2 * POWER_CHUNK_SAMPLES):
    """5th-percentile floor, +rise_db threshold, contiguous runs above it as
    (start_byte, end_byte) with end exclusive in chunks (worker.py:241-264).
    Returns (baseline, threshold_linear, ranges)."""
    pm = np.asarray(power_map)
    if pm.size == 0:
        return 0.0, 0.0, []
    baseline = np.percentile(pm, 5)                          # worker.py:242
    if baseline <= 0:
        baseline = 1.0                                       # worker.py:243
    thr = baseline * 10 ** (rise_db / 10.0)                  # worker.py:245-246
    mask = pm > thr
    ranges = []
    if mask.any():
        edges = np.diff(mask.astype(int))                    # worker.py:254
        starts = np.where(edges == 1)[0] + 1
        ends = np.where(edges == -1)[0] + 1
        if mask[0]:
            starts = np.insert(starts, 0, 0)
        if mask[-1]:
            ends = np.append(ends, pm.size)
        ranges = [(s * chunk_bytes, e * chunk_bytes) for s, e in zip(starts, ends)]
    return baseline, thr, ranges


# ----------------------------------------------------------------------------------
# (1b) checkIfJamming flavour of the same scan    checkIfJamming.py:7-106
# ----------------------------------------------------------------------------------
def cij_chunk_power(piece: np.ndarray, threshold: float):
    """(is_jamming, avg_power) of one chunk; odd-sized or empty chunk -> (False, 0.0)
    (checkIfJamming.py:12-20).  Uses |complex64|^2, not i^2+q^2."""
    piece = _as_u8(piece)
    if piece.size % 2 != 0 or piece.size == 0:
        return False, 0.0
    f = piece.astype(np.float32) - 127.5
    z = f[0::2] + 1j * f[1::2]
    avg = np.mean(np.abs(z) ** 2)
    return bool(avg > threshold), avg


def cij_events(raw, threshold: float, chunk_bytes: int = CIJ_CHUNK_BYTES):
    """List of (start_sample, end_sample) runs above ``threshold``
    (checkIfJamming.py:22-63)."""
    raw = _as_u8(raw)
    events, active, start, done = [], False, None, 0
    for off in range(0, raw.size, chunk_bytes):
        piece = raw[off:off + chunk_bytes]
        n_new = piece.size // 2
        if n_new == 0:
            continue                                         # checkIfJamming.py:38-39
        hot, _ = cij_chunk_power(piece, threshold)
        if hot and not active:
            start = done
        elif not hot and active and start is not None:
            events.append((start, done))
            start = None
        active = hot
        done += n_new
    if active and start is not None:
        events.append((start, done))
    return events


def cij_calibrate(raw, chunk_bytes: int = CIJ_CHUNK_BYTES):
    """(median, max, min, suggested_threshold) of the per-chunk powers
    (checkIfJamming.py:69-103); None for an empty stream."""
    raw = _as_u8(raw)
    powers = [cij_chunk_power(raw[o:o + chunk_bytes], 0.0)[1]
              for o in range(0, raw.size, chunk_bytes)]
    if not powers:
        return None
    p = np.array(powers)
    med = np.median(p)
    return med, np.max(p), np.min(p), med * CIJ_CALIBRATION_FACTOR


# ----------------------------------------------------------------------------------
# (2) RSSI amplitude statistics -> distance -> grid search    triangulateRSSI.py
# ----------------------------------------------------------------------------------
def rssi_unpack(raw) -> np.ndarray:
    """(u8 - 127.5)/127.5 -> complex64 (triangulateRSSI.py:29-31)."""
    raw = _as_u8(raw)
    f = (raw.astype(np.float32) - 127.5) / 127.5
    return f[0::2] + 1j * f[1::2]


def rssi_amp_stats(raw, threshold: float):
    """(first_index_over_threshold | None, mean amplitude from that index on as
    np.float32 | None)  -- triangulateRSSI.py:37-40,65-68."""
    z = rssi_unpack(raw)
    if z.size == 0:
        return None, None
    amp = np.abs(z)
    hits = np.where(amp > threshold)[0]
    if hits.size == 0:
        return None, None
    k = int(hits[0])
    return k, np.mean(amp[k:])


def rssi_distance(raw, tx_power=RSSI_TX_POWER, path_loss_exp=RSSI_PATH_LOSS_EXP,
                  frequency_mhz=RSSI_FREQ_MHZ, threshold=RSSI_THRESHOLD):
    """Log-distance range estimate from the mean amplitude
    (triangulateRSSI.py:54-82).  None when nothing exceeds the threshold."""
    k, avg = rssi_amp_stats(raw, threshold)
    if k is None or avg == 0:
        return None
    return distance_from_mean_amplitude(avg, tx_power, path_loss_exp, frequency_mhz)


def distance_from_mean_amplitude(avg, tx_power=RSSI_TX_POWER,
                                 path_loss_exp=RSSI_PATH_LOSS_EXP,
                                 frequency_mhz=RSSI_FREQ_MHZ):
    """triangulateRSSI.py:70-75 with the reference's dtype flow: ``avg`` is float32,
    so 10*log10(avg**2) is float32; the 1 m path loss is float64."""
    avg = np.float32(avg)
    prx_db = 10 * np.log10(avg ** 2)
    pl_1m = 20 * np.log10(frequency_mhz) - 27.55
    return 10 ** ((tx_power - prx_db - pl_1m) / (10 * path_loss_exp))


def grid_search(positions, radii, density: int = GRID_DENSITY,
                range_mult: float = GRID_RANGE_MULT) -> np.ndarray:
    """argmin over a density x density grid of sum_k | |p - a_k| - r_k |
    (triangulateRSSI.py:88-120), first minimum in row-major order."""
    pos = np.array(positions)
    rad = np.array(radii)
    half = np.max(rad) * range_mult
    c = np.mean(pos, axis=0)
    xs = np.linspace(c[0] - half, c[0] + half, density)
    ys = np.linspace(c[1] - half, c[1] + half, density)
    gx, gy = np.meshgrid(xs, ys)
    err = np.zeros_like(gx)
    for a, r in zip(pos, rad):
        err += np.abs(np.sqrt((gx - a[0]) ** 2 + (gy - a[1]) ** 2) - r)
    iy, ix = np.unravel_index(np.argmin(err), err.shape)
    return np.array([gx[iy, ix], gy[iy, ix]])


def meters_to_degrees(mx, my, ref_lat=50.0):
    """triangulateRSSI.py:42-52."""
    dlat = my / METERS_PER_DEG
    dlon = mx / (METERS_PER_DEG * math.cos(math.radians(ref_lat)))
    return dlat, dlon, dlat * 60, dlon * 60


def triangulate(raws, antenna_positions_meters=None, reference_lat=50.00898,
                reference_lon=19.98287, tx_power=RSSI_TX_POWER,
                path_loss_exp=RSSI_PATH_LOSS_EXP, frequency_mhz=RSSI_FREQ_MHZ,
                threshold=RSSI_THRESHOLD):
    """Result dict of triangulate_jammer_location (triangulateRSSI.py:126-229) for
    in-memory streams (``raws``: list of uint8 arrays; None = missing file)."""
    n_files = len(raws)
    if n_files < 2:
        return dict(success=False, distances=None, location_meters=None,
                    location_geographic=None,
                    message='Wymagane są co najmniej 2 pliki z danymi anten.',
                    num_antennas=n_files)
    if antenna_positions_meters is None:
        antenna_positions_meters = [np.array([0.0, 0.0]), np.array([0.5, 0.0]),
                                    np.array([0.0, 0.5])][:n_files]
    dists, use_pos, use_r = [], [], []
    for k, raw in enumerate(raws):
        d = None if raw is None else rssi_distance(raw, tx_power, path_loss_exp,
                                                   frequency_mhz, threshold)
        dists.append(d)
        if d is not None and k < len(antenna_positions_meters):
            use_r.append(d)
            use_pos.append(np.array(antenna_positions_meters[k]))
    if len(use_r) < 2:
        return dict(success=False, distances=dists, location_meters=None,
                    location_geographic=None,
                    message='Nie udało się obliczyć poprawnej odległości dla '
                            'wystarczającej liczby anten (min 2). Sukcesy: '
                            f'{len(use_r)}',
                    num_antennas=n_files)
    best = grid_search(use_pos, use_r)
    dlat, dlon, dlat_m, dlon_m = meters_to_degrees(best[0], best[1], reference_lat)
    return dict(
        success=True, distances=dists, location_meters=best.tolist(),
        location_geographic=dict(lat=reference_lat + dlat, lon=reference_lon + dlon,
                                 lat_offset_degrees=dlat, lon_offset_degrees=dlon,
                                 lat_offset_minutes=dlat_m, lon_offset_minutes=dlon_m),
        message='Lokalizacja wyznaczona algorytmem Grid Search (błąd minimalny). '
                f'x={best[0]:.2f}m, y={best[1]:.2f}m',
        num_antennas=len(use_r))


# ----------------------------------------------------------------------------------
# (3) Welch PSD waterfall     widmo_plot.py:26-54 + scipy.signal.welch
# ----------------------------------------------------------------------------------
def hann_periodic(n: int) -> np.ndarray:
    """scipy.signal.get_window('hann', n) (fftbins=True): 0.5 - 0.5 cos(2 pi k / n)."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def welch_twosided_c64(x: np.ndarray, fs: float, nperseg: int) -> np.ndarray:
    """scipy.signal.welch(x, fs, nperseg=nperseg, return_onesided=False) for complex64
    ``x`` with the defaults the reference relies on: periodic Hann, noverlap =
    nperseg//2, nfft = nperseg, detrend='constant' per segment, scaling='density',
    average='mean', no padding/boundary (scipy/signal/_spectral_py.py
    _spectral_helper/_fft_helper, scipy 1.15.3).  Returns float32[nperseg]."""
    x = np.asarray(x)
    if x.size < nperseg:                       # _triage_segments: shrink to input length
        nperseg = x.size
    win = hann_periodic(nperseg)
    scale = 1.0 / (fs * (win * win).sum())
    win_c = win.astype(np.result_type(x, np.complex64))
    step = nperseg - nperseg // 2
    segs = np.lib.stride_tricks.sliding_window_view(x, nperseg)[0::step]
    segs = segs - np.mean(segs, axis=-1, keepdims=True)      # detrend 'constant'
    spec = _sfft.fft(win_c * segs, n=nperseg)
    pxx = np.conjugate(spec) * spec
    pxx *= scale
    pxx = pxx.astype(spec.dtype).real
    if pxx.shape[0] > 1:
        pxx = pxx.mean(axis=0)
    else:
        pxx = pxx[0]
    return pxx


def widmo_chunk_psd_db(raw_chunk, fs: float = SAMPLE_RATE,
                       nperseg: int = WELCH_NPERSEG):
    """One waterfall row (widmo_plot.py:38-52): returns (psd_linear_shifted float32,
    psd_db float32)."""
    raw_chunk = _as_u8(raw_chunk)
    f = raw_chunk.astype(np.float32)
    z = (f[0::2] - 127.5) / 127.5 + 1j * ((f[1::2] - 127.5) / 127.5)
    z = z - np.mean(z)                                        # widmo_plot.py:44
    p = np.fft.fftshift(welch_twosided_c64(z, fs, nperseg))   # widmo_plot.py:48,51
    return p, 10 * np.log10(p + 1e-15)                        # widmo_plot.py:52


def widmo_waterfall(raw, fs: float = SAMPLE_RATE, nperseg: int = WELCH_NPERSEG,
                    chunk_samples: int | None = None):
    """All rows of the waterfall + byte samples for the histogram
    (widmo_plot.py:26-57).  Returns (psd_lin[rows, nperseg], psd_db[rows, nperseg],
    hist_samples uint8)."""
    raw = _as_u8(raw)
    if chunk_samples is None:
        chunk_samples = int(fs)                               # widmo_plot.py:9
    step = 2 * chunk_samples
    lin, db, hist = [], [], []
    for off in range(0, raw.size, step):
        piece = raw[off:off + step]
        if piece.size < 2 * nperseg:                          # widmo_plot.py:31
            break
        hist.append(piece[::100])                             # widmo_plot.py:35
        p, d = widmo_chunk_psd_db(piece, fs, nperseg)
        lin.append(p)
        db.append(d)
    if not lin:
        return (np.zeros((0, nperseg), np.float32), np.zeros((0, nperseg), np.float32),
                np.zeros(0, np.uint8))
    return np.array(lin), np.array(db), np.concatenate(hist)


# ----------------------------------------------------------------------------------
# (4) TDOA: onset detection + FFT cross-correlation lag     triangulateTDOA.py
# ----------------------------------------------------------------------------------
def tdoa_unpack(raw) -> np.ndarray:
    """(I-127.5) + j(Q-127.5), un-normalised complex64 (triangulateTDOA.py:33-34)."""
    raw = _as_u8(raw)
    return (raw[0::2].astype(np.float32) - 127.5) + 1j * (raw[1::2].astype(np.float32) - 127.5)


def tdoa_onset(z: np.ndarray, noise_samples: int = TDOA_NOISE_SAMPLES,
               window: int = TDOA_WINDOW, factor: float = TDOA_FACTOR) -> int:
    """First index where the ``window``-tap moving average of |z|^2 exceeds
    ``factor`` x mean(|z[:noise_samples]|^2), plus window//2; -1 if none / too short
    (triangulateTDOA.py:37-49)."""
    if len(z) < noise_samples + window:
        return -1
    p = np.abs(z) ** 2
    noise = np.mean(p[:noise_samples])
    if noise == 0:
        noise = 1e-9
    ma = np.convolve(p, np.ones(window) / window, mode='valid')
    hits = np.where(ma > noise * factor)[0]
    return int(hits[0]) + window // 2 if hits.size else -1


def xcorr_full_fft(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """scipy.signal.correlate(a, b, mode='full') through its FFT branch:
    fftconvolve(a, conj(b[::-1])) with FFT length next_fast_len(len(a)+len(b)-1)
    (scipy/signal/_signaltools.py correlate/_freq_domain_conv, scipy 1.15.3)."""
    a = np.asarray(a)
    b = np.asarray(b)
    br = b[::-1].conj()
    full = a.size + br.size - 1
    nfft = _sfft.next_fast_len(full, False)
    prod = _sfft.fft(a, nfft) * _sfft.fft(br, nfft)
    return _sfft.ifft(prod, nfft)[:full]


def xcorr_lag(sig1: np.ndarray, sig0: np.ndarray):
    """(lag, |c|max) with lag = argmax|correlate(sig1, sig0,'full')| - (len(sig0)-1)
    (triangulateTDOA.py:86-89); positive lag = sig1 delayed w.r.t. sig0."""
    c = np.abs(xcorr_full_fft(sig1, sig0))
    k = int(np.argmax(c))
    return k - (len(sig0) - 1), c[k]


def tdoa_bearing(lag: int, ant0, ant1, fs: float = 2048000):
    """Scalar geometry of triangulateTDOA.py:92-119, including the reference's
    atan2(dy, 0) baseline-angle quirk (:114).  Returns dict or None when
    |cos| > 1 / zero baseline."""
    ant0 = np.asarray(ant0, dtype=float)
    ant1 = np.asarray(ant1, dtype=float)
    tdoa = lag / fs
    path = tdoa * SPEED_OF_LIGHT
    base = float(np.linalg.norm(ant1 - ant0))
    if base == 0:
        return None
    c = path / base
    if abs(c) > 1:
        return None
    theta = math.acos(c)
    base_ang = math.atan2(ant1[1] - ant0[1], ant0[0] - ant0[0])
    return dict(tdoa=tdoa, path_difference=path, antenna_distance=base,
                theta_deg=math.degrees(theta),
                azimuth1_deg=math.degrees(base_ang + theta) % 360,
                azimuth2_deg=math.degrees(base_ang - theta) % 360)


# ----------------------------------------------------------------------------------
# (6) GNSS acquisition search      GpsJammerApp/backend/sdracq.c, sdrcmn.c
# ----------------------------------------------------------------------------------
# PARITY UNPINNED for everything behind the first FFT: the reference's acquisition lives in
# gnssdec (C + FFTW3f), which cannot be built here (fftw3.h / fec.h / libusb.h are absent)
# and whose shipped binary is never run, and the reference holds no test vector for it.  What IS
# pinned: the C/A codes, against the published first-ten-chips table of IS-GPS-200 (Table 3-Ia) in
# tests/test_acq_host.py.  The functions below restate the C line by line in numpy (float32 where the
# C uses float, the FFT by scipy.fft / pocketfft in complex64 where the C calls FFTW).
ACQ_CSCALE = 1.0 / 32.0              # sdrcmn.c:7
ACQ_TH = 3.0                         # sdr.h:66
ACQ_G2_DELAY = (5, 6, 7, 8, 17, 18, 139, 140, 141, 251, 252, 254, 255, 256, 257, 258, 469, 470, 471,
                472, 473, 474, 509, 512, 513, 514, 515, 516, 859, 860, 861, 862)   # sdrcode.c:104-106 (PRN 1..32)


def acq_gencode_l1ca(prn: int) -> np.ndarray:
    """gencode_L1CA (sdrcode.c:102-149): registers of -1, products for XOR."""
    R1 = [-1] * 10
    R2 = [-1] * 10
    G1 = np.empty(1023, np.int16)
    G2 = np.empty(1023, np.int16)
    for i in range(1023):
        G1[i] = R1[9]
        G2[i] = R2[9]
        C1 = R1[2] * R1[9]
        C2 = R2[1] * R2[2] * R2[5] * R2[7] * R2[8] * R2[9]
        for j in range(9, 0, -1):
            R1[j] = R1[j - 1]
            R2[j] = R2[j - 1]
        R1[0] = C1
        R2[0] = C2
    code = np.empty(1023, np.int16)
    j = 1023 - ACQ_G2_DELAY[prn - 1]
    for i in range(1023):
        code[i] = -G1[i] * G2[j % 1023]
        j += 1
    return code


def acq_rescode(code: np.ndarray, ci: float, n: int) -> np.ndarray:
    """rescode(code, len, 0, 0, ci, n, rcode), SSE2 form (sdrcmn.c:541-575), one sample at a time."""
    ln = len(code)
    i, nbit = ln, 31
    while i:
        i >>= 1
        nbit -= 1
    nbit -= 1
    scale = 1 << nbit
    coff = 0.0
    x = []
    for _ in range(4):
        x.append(int(coff * scale + 0.5))
        coff += ci
    step = int(ci * 4 * scale + 0.5)
    out = np.empty(n, np.int16)
    for g in range(0, n, 4):
        for k in range(4):
            if x[k] > ln * scale - 1:
                x[k] -= ln * scale
            if g + k < n:
                out[g + k] = code[x[k] >> nbit]
            x[k] += step
    return out


def acq_mixcarr_sse2(data_i8: np.ndarray, ti: float, n: int, freq: float):
    """mixcarr(data, DTYPEIQ, ti, n, freq, 0.0, II, QQ), SSE2 form (sdrcmn.c:618-684): 16-entry int8 table,
    sixteen double phases advanced by 16 ps per block, cvttpd + mask."""
    cost = np.array([math.floor(math.cos(2 * math.pi / 16 * i) / ACQ_CSCALE + 0.5) for i in range(16)], np.int32)
    sint = np.array([math.floor(math.sin(2 * math.pi / 16 * i) / ACQ_CSCALE + 0.5) for i in range(16)], np.int32)
    ps = freq * 16 * ti
    phi = 0.0
    regs = []
    for _ in range(8):                                   # xmm1..xmm8 = (phi, phi + ps); phi += ps * 2
        regs += [phi, phi + ps]
        phi += ps * 2
    regs = np.array(regs, np.float64)
    inc = ps * 16
    d = data_i8.astype(np.int32)
    II = np.empty(n, np.int16)
    QQ = np.empty(n, np.int16)
    for b in range(0, n, 16):
        idx = np.trunc(regs).astype(np.int64) & 15
        di, dq = d[2 * b:2 * b + 32:2], d[2 * b + 1:2 * b + 32:2]
        II[b:b + 16] = cost[idx] * di - sint[idx] * dq
        QQ[b:b + 16] = sint[idx] * di + cost[idx] * dq
        regs = regs + inc
    return II, QQ


def acq_code_fft(prn: int, nsamp: int, fs: float = SAMPLE_RATE) -> np.ndarray:
    """xcode of a channel (sdrinit.c:436-441): resampled code, zero-padded to nfft, forward FFT."""
    ci = (1.0 / fs) * 1.023e6
    rcode = np.zeros(2 * nsamp, np.int16)
    rcode[:nsamp] = acq_rescode(acq_gencode_l1ca(prn), ci, nsamp)
    return _sfft.fft(rcode.astype(np.float32).astype(np.complex64))


def acq_pcorrelator(data_i8: np.ndarray, ti: float, n: int, freqs, m: int, codex: np.ndarray, P: np.ndarray):
    """pcorrelator (sdrcmn.c:742-773) + cpxconv (:124-147): P[i*n + k] += |IFFT(-FFT(x_i) conj-product)|^2 / m^2."""
    m2 = np.float32(m) * np.float32(m)
    for i, freq in enumerate(freqs):
        II, QQ = acq_mixcarr_sse2(data_i8, ti, m, float(freq))
        sc = np.float32(ACQ_CSCALE / m)
        x = (II.astype(np.float32) * sc + 1j * (QQ.astype(np.float32) * sc)).astype(np.complex64)   # cpxcpx
        X = _sfft.fft(x)
        p0, p1, q0, q1 = X.real, X.imag, codex.real, codex.imag
        Y = ((-p0 * q0 - p1 * q1) + 1j * (p0 * q1 - p1 * q0)).astype(np.complex64)
        y = _sfft.ifft(Y) * np.float32(m)                    # FFTW's backward transform is unnormalised
        y = y.astype(np.complex64)
        term = (y.real[:n] * y.real[:n] + y.imag[:n] * y.imag[:n]) / m2
        P[i * n:(i + 1) * n] += term.astype(np.float64)


def _acq_kept(i, exinds, exinde):
    return (i < exinds or i > exinde) if exinds <= exinde else (i < exinds and i > exinde)


def acq_check(P: np.ndarray, nsamp: int, nfreq: int, nsampchip: int, ctime: float, th: float = ACQ_TH):
    """checkacquisition (sdracq.c:52-84) with maxvd / meanvd / ind2sub as written (sdrcmn.c:411-440,512-515)."""
    maxi = int(np.argmax(P))                              # maxvd without exclusion: first maximum
    maxP = float(P[maxi])
    codei, freqi = maxi % nsamp, (nfreq * maxi) // (nsamp * nfreq)
    exinds = codei - 2 * nsampchip
    if exinds < 0:
        exinds += nsamp
    exinde = codei + 2 * nsampchip
    if exinde >= nsamp:
        exinde -= nsamp
    row = P[freqi * nsamp:(freqi + 1) * nsamp]
    keep = np.array([_acq_kept(i, exinds, exinde) for i in range(nsamp)])
    meanP = float(row[keep].sum() / keep.sum())
    mx = float(row[0])                                    # maxvd seeds with data[0] whatever the exclusion zone
    for i in range(1, nsamp):
        if keep[i] and mx < row[i]:
            mx = float(row[i])
    return {"maxP": maxP, "maxP2": mx, "meanP": meanP, "peakr": maxP / mx,
            "cn0": 10 * math.log10(maxP / meanP / ctime), "codei": codei, "freqi": freqi,
            "acquired": (maxP / mx) > th}


def acq_search(raw, first_sample: int, prn: int, fs: float = SAMPLE_RATE, f_if: float = 0.0, intg: int = 10,
               hband: int = 7000, step: int = 200):
    """sdraqcuisition for one channel (sdracq.c:3-50) on the uint8 capture: int8 = u8 - 128 (sdrrcv.c:104-106),
    up to ``intg`` steps of 2*nsamp samples advancing by nsamp, stop at the first step that acquires.
    Returns (check dict + 'steps', P[nfreq*nsamp])."""
    raw = _as_u8(raw)
    ctime = 1023 / 1.023e6
    nsamp = int(fs * ctime)
    nsampchip = int(nsamp / 1023)
    nfreq = 2 * (hband // step) + 1
    freqs = [f_if + (i - (nfreq - 1) // 2) * step for i in range(nfreq)]
    codex = acq_code_fft(prn, nsamp, fs)
    P = np.zeros(nfreq * nsamp, np.float64)               # calloc (sdrmain.c:346)
    res = None
    for i in range(intg):
        loc = first_sample + i * nsamp
        win = raw[2 * loc:2 * (loc + 2 * nsamp)]
        data = (win.astype(np.int16) - 128).astype(np.int8)
        acq_pcorrelator(data, 1.0 / fs, nsamp, freqs, 2 * nsamp, codex, P)
        res = acq_check(P, nsamp, nfreq, nsampchip, ctime)
        res["steps"] = i + 1
        if res["acquired"]:
            break
    return res, P
