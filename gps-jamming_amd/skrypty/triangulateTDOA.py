"""Drop-in for skrypty/triangulateTDOA.py of mfkiwl/GPS-JAMMING, MI355X-backed.

Two-antenna TDOA bearing: software synchronisation on the power onset of each capture, FFT
cross-correlation of onset-aligned slices, lag -> path difference -> angle of arrival
(reference: skrypty/triangulateTDOA.py:51-127).  Same constants, same function names, same
console report when run as a script.

On the GPU (libgpsjam_hip.so): ``find_interference_start`` (K4, reference :37-49) and the
slice correlation + arg-max (K5, reference :80-89).  Both work on the raw uint8 captures; the
complex64 expansion of the whole file that the reference performs (:33-34) never happens.
``load_iq_data`` therefore returns a light handle that keeps the bytes; code that indexes it
like an array still gets the reference's un-normalised complex64 values.
"""
import logging
import math
import os
import sys

import numpy as np

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # .../gps-jamming_amd
if _PKG_ROOT not in sys.path:
    sys.path.append(_PKG_ROOT)

import gpsjam   # noqa: E402

# --- configuration (reference :9-29) --------------------------------------------------------
FILE_ANT0 = '17_10/capture1710_0_15m.bin'
FILE_ANT1 = '17_10/capture1710_1.bin'
SAMPLE_RATE = 2048000
CENTER_FREQ = 1575420000
ANT0_POS = np.array([0, 0])
ANT1_POS = np.array([0.5, 0])
NOISE_SAMPLE_SIZE = 200000
DETECTION_WINDOW_SIZE = 1000
DETECTION_THRESHOLD_FACTOR = 50.0
CORRELATION_SLICE_SIZE = 50000
SPEED_OF_LIGHT = 299792458


class IQCapture:
    """uint8 capture with the array surface the reference script uses on its complex
    arrays: ``len()``, slicing (-> IQCapture) and ``np.asarray`` (-> complex64 of
    (I-127.5) + j(Q-127.5), reference :34)."""

    def __init__(self, raw, path=None):
        raw = gpsjam.as_u8(raw)
        self.raw = raw[:raw.size - (raw.size & 1)]
        # the capture FILE these bytes are (load_iq_data; None for slices and arrays): lets the onset search run on the
        # process-wide resident copy of the file (gpsjam.resident_capture) instead of uploading the bytes again
        self.path = path

    def __len__(self):
        return self.raw.size // 2

    def __getitem__(self, key):
        if isinstance(key, slice):
            start, stop, step = key.indices(len(self))
            if step != 1:
                raise IndexError("IQCapture supports contiguous slices only")
            return IQCapture(self.raw[2 * start:2 * max(stop, start)])
        return np.asarray(self)[key]

    def __array__(self, dtype=None, copy=None):
        z = (self.raw[0::2].astype(np.float32) - 127.5) + 1j * (self.raw[1::2].astype(np.float32) - 127.5)
        return z if dtype is None else z.astype(dtype)


def load_iq_data(filename):
    """Capture handle (reference :31-35 returned the expanded complex64 array)."""
    return IQCapture(gpsjam.read_capture(filename), path=filename)


def _raw_of(iq_data):
    return iq_data.raw if isinstance(iq_data, IQCapture) else None


# Rounding bands inside which the GPU's decision and the reference's may differ: the kernels sum
# exact integers / run a radix-16 complex64 FFT, the reference sums float32 |z|^2 (pairwise mean,
# float64 inside np.convolve) / runs pocketfft.  Either side carries ~3e-7 (onset threshold) and
# ~1e-6 (correlation peak) of rounding.  K4 reports the first index whose exact moving average is
# inside (or above) the band, ``guard_index``: in front of it the reference cannot cross, so when a
# decision does fall inside the band the reference's own expression is evaluated on the host FROM
# THERE ON (a few windows, not the capture) and decides; outside the band the index is the
# reference's by construction.
ONSET_NEAR_TIE = 1e-6
LAG_NEAR_TIE = 2e-5
near_tie_events = []     # (what, margin): filled when the host re-evaluation ran (diagnostics / tests)
_log = logging.getLogger("gpsjam.tdoa")


def _onset_reference_expression(raw, noise_samples, window_size, threshold_factor, first_position=0,
                                fetch=None, n_samples=None):
    """The reference's own arithmetic (:37-49) on the raw bytes: float32 |z|^2, float32 mean of the
    noise span, float64 moving average by np.convolve(..., 'valid'), first index above the threshold.
    Evaluated from moving-average position ``first_position`` on, in blocks that grow from a few
    windows to 4 Mi positions, so that the usual near-tie (the reference crosses within a window or
    two of the exact crossing) costs microseconds and a capture is never expanded 8x at once.
    ``fetch(a, b)`` returns the uint8 bytes of samples [a, b) (a resident capture); default: ``raw``."""
    n = (raw.size // 2) if n_samples is None else int(n_samples)
    if fetch is None:
        def fetch(a, b):
            return raw[2 * a:2 * b]

    def power_of(a, b):
        seg = fetch(a, b)
        z = (seg[0::2].astype(np.float32) - 127.5) + 1j * (seg[1::2].astype(np.float32) - 127.5)
        return np.abs(z.astype(np.complex64)) ** 2

    noise_power = np.mean(power_of(0, noise_samples))
    if noise_power == 0:
        noise_power = 1e-9
    threshold = noise_power * threshold_factor
    kernel = np.ones(window_size) / window_size
    n_out = n - window_size + 1
    o0 = max(int(first_position), 0)
    block = max(4 * window_size, 1 << 12)
    while o0 < n_out:
        o1 = min(o0 + block, n_out)
        ma = np.convolve(power_of(o0, o1 + window_size - 1), kernel, mode='valid')
        hit = np.where(ma > threshold)[0]
        if hit.size:
            return int(o0 + hit[0] + window_size // 2)
        o0, block = o1, min(4 * block, 1 << 22)
    return -1


def _resolve_near_tie(res, window_size, evaluate):
    """K4's result -> the reference's index.  ``evaluate(first_position)`` runs the reference expression."""
    if not res.near_tie:
        return int(res.start_index)
    near_tie_events.append(("onset", float(res.margin)))
    if res.guard_index < 0:        # nothing reaches even the band: cannot happen with near_tie, kept for safety
        return -1
    first = int(res.guard_index) - int(window_size) // 2
    got = evaluate(first)
    _log.warning("onset decided inside the rounding band (margin %.2e at index %d, band from %d): the reference's "
                 "float32 expression was evaluated on the host from there on -> %d",
                 res.margin, res.start_index, res.guard_index, got)
    return got


def find_interference_start(iq_data, noise_samples, window_size, threshold_factor):
    """Sample index where the moving-average power first exceeds threshold_factor x the
    noise power, + window_size // 2; -1 if none (reference :37-49)."""
    raw = _raw_of(iq_data)
    if raw is None:
        raise TypeError("find_interference_start expects the capture returned by load_iq_data")
    if len(iq_data) < noise_samples + window_size:
        return -1
    dev = gpsjam.default_device()
    src = raw
    if getattr(iq_data, "path", None) and raw.size:
        # file-backed: one upload per file and process, shared with the worker's scan and the RSSI solver; on first
        # use the onset is computed while the file uploads
        try:
            src = gpsjam.resident_capture(iq_data.path, noise_samples=int(noise_samples), window=int(window_size),
                                          factor=float(threshold_factor))
            if src.nbytes != raw.size + (src.nbytes & 1):     # the file changed under the handle: use the bytes we hold
                src = raw
        except OSError:
            src = raw
    res = dev.onset(src, int(noise_samples), int(window_size), float(threshold_factor))
    return _resolve_near_tie(res, window_size, lambda first: _onset_reference_expression(
        raw, int(noise_samples), int(window_size), float(threshold_factor), first))


def correlation_lag(signal1_slice, signal0_slice):
    """argmax |correlate(signal1, signal0, 'full')| - (len(signal0) - 1) (reference :86-89),
    positive when antenna 1 receives the signal later.  Returns (lag, peak)."""
    r1, r0 = _raw_of(signal1_slice), _raw_of(signal0_slice)
    if r1 is None or r0 is None or r1.size != r0.size:
        raise TypeError("correlation_lag expects two equal-length slices of load_iq_data captures")
    lags, peaks, margins = gpsjam.default_device().xcorr_lags([r0, r1], [(0, 1)], want_margins=True)
    if margins[0] >= LAG_NEAR_TIE:
        return int(lags[0]), float(peaks[0])
    # two lags within FFT rounding of each other: the reference's choice depends on ITS rounding,
    # so its own call (:86-89) decides
    near_tie_events.append(("lag", float(margins[0])))
    _log.warning("lag decided inside the rounding of a complex64 FFT (peak margin %.2e): scipy.signal.correlate "
                 "evaluated on the host", float(margins[0]))
    from scipy import signal
    corr = signal.correlate(np.asarray(signal1_slice), np.asarray(signal0_slice), mode='full')
    k = int(np.argmax(np.abs(corr)))
    return k - (len(signal0_slice) - 1), float(np.abs(corr[k]))


def bearing_from_lag(lag_samples, ant0_pos=ANT0_POS, ant1_pos=ANT1_POS, sample_rate=SAMPLE_RATE):
    """Scalar geometry of reference :92-119 (including its baseline-angle expression, :114).
    Returns a dict, or a dict with 'error' when the geometry is impossible."""
    tdoa = lag_samples / sample_rate
    path_difference = tdoa * SPEED_OF_LIGHT
    antenna_distance = np.linalg.norm(np.asarray(ant1_pos) - np.asarray(ant0_pos))
    out = {'tdoa': tdoa, 'path_difference': path_difference, 'antenna_distance': antenna_distance}
    if antenna_distance == 0:
        out['error'] = 'zero_baseline'
        return out
    cos_theta = path_difference / antenna_distance
    if abs(cos_theta) > 1:
        out['error'] = 'path_longer_than_baseline'
        return out
    theta = math.acos(cos_theta)
    baseline = math.atan2(ant1_pos[1] - ant0_pos[1], ant0_pos[0] - ant0_pos[0])
    out.update(theta_deg=math.degrees(theta),
               azimuth1_deg=math.degrees(baseline + theta) % 360,
               azimuth2_deg=math.degrees(baseline - theta) % 360)
    return out


def main(file0=FILE_ANT0, file1=FILE_ANT1):
    print("Wczytywanie danych I/Q...")
    try:
        signal0_full = load_iq_data(file0)
        signal1_full = load_iq_data(file1)
    except FileNotFoundError as e:
        print(f"Błąd: Nie znaleziono pliku! {e}")
        return 1

    print("\nRozpoczynanie synchronizacji programowej...")
    start0 = find_interference_start(signal0_full, NOISE_SAMPLE_SIZE, DETECTION_WINDOW_SIZE, DETECTION_THRESHOLD_FACTOR)
    start1 = find_interference_start(signal1_full, NOISE_SAMPLE_SIZE, DETECTION_WINDOW_SIZE, DETECTION_THRESHOLD_FACTOR)
    if start0 == -1 or start1 == -1:
        print("BŁĄD KRYTYCZNY: Nie udało się wykryć początku interferencji.")
        return 1
    print(f"Wykryto początek interferencji w pliku 0 na próbce: {start0}")
    print(f"Wykryto początek interferencji w pliku 1 na próbce: {start1}")

    if len(signal0_full) < start0 + CORRELATION_SLICE_SIZE or len(signal1_full) < start1 + CORRELATION_SLICE_SIZE:
        print("BŁĄD: Niewystarczająca ilość danych po wykryciu interferencji do analizy.")
        return 1
    signal0_slice = signal0_full[start0:start0 + CORRELATION_SLICE_SIZE]
    signal1_slice = signal1_full[start1:start1 + CORRELATION_SLICE_SIZE]
    print(f"\nSygnały wyrównane. Przetwarzanie wycinka {CORRELATION_SLICE_SIZE} próbek.")

    print("Obliczanie korelacji wzajemnej na wycinkach sygnału...")
    lag_samples, _ = correlation_lag(signal1_slice, signal0_slice)
    print(f"Znaleziono maksymalną korelację przy przesunięciu {lag_samples} próbek.")

    geo = bearing_from_lag(lag_samples)
    print(f"Różnica czasu dotarcia (TDOA): {geo['tdoa'] * 1e9:.2f} ns")
    print(f"Różnica w odległości do anten: {geo['path_difference']:.4f} m")
    if geo.get('error') == 'zero_baseline':
        print("Błąd: Odległość między antenami wynosi 0.")
        return 1
    if geo.get('error'):
        print("\nOSTRZEŻENIE: Obliczona różnica ścieżek jest większa niż odległość między antenami.")
        print("Możliwe przyczyny: błąd w konfiguracji odległości anten lub bardzo silne odbicia (multipath).")
        return 1

    print("\n--- WYNIKI ---")
    print(f"Odległość między antenami: {geo['antenna_distance']:.2f} m")
    print(f"Kąt nadejścia fali interferencyjnej (względem osi anten): {geo['theta_deg']:.2f} stopni")
    print("Potencjalne kierunki do źródła interferencji (azymuty):")
    print(f"  Kierunek 1: {geo['azimuth1_deg']:.2f} stopni")
    print(f"  Kierunek 2: {geo['azimuth2_deg']:.2f} stopni")
    return 0


if __name__ == "__main__":
    sys.exit(main(*sys.argv[1:3]))
