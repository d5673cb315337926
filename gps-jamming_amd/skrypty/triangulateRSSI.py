"""Drop-in for skrypty/triangulateRSSI.py of mfkiwl/GPS-JAMMING, MI355X-backed.

Same module name, same public names, same ``triangulate_jammer_location`` signature and
result dict (reference: skrypty/triangulateRSSI.py:126-229), so ``GpsJammerApp/app/worker.py``
(which does ``from triangulateRSSI import triangulate_jammer_location`` after putting this
directory on sys.path, worker.py:13-14) and the GUI consume it unchanged.

What moved to the GPU: the whole-file pass of ``calculate_distance_from_file`` -- unpack,
|x|, first index above the threshold, mean amplitude from there (reference :29-31,:65-68) --
is one call into libgpsjam_hip.so (kernel K3, ``gj_amp_stats_u8``) on the memory-mapped
capture; nothing of the file is expanded on the host.  The scalar tail (dB, log-distance
model) and the 300 x 300 grid search stay on the host in numpy with the reference's dtype
flow, so ``location_meters`` is bit-identical for identical distances.

``read_iq_data`` / ``find_change_point`` are kept for callers that import them; they operate
on host arrays by definition and are not used by the path above.
"""
import math
import os
import sys

import numpy as np

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # .../gps-jamming_amd
if _PKG_ROOT not in sys.path:
    sys.path.append(_PKG_ROOT)

import gpsjam   # noqa: E402

# --- calibration defaults (reference :9-12) -------------------------------------------------
DEFAULT_CALIBRATED_TX_POWER = 40.0
DEFAULT_CALIBRATED_PATH_LOSS_EXPONENT = 3.0
DEFAULT_SIGNAL_FREQUENCY_MHZ = 1575.42
DEFAULT_SIGNAL_THRESHOLD = 0.1

# --- grid search (reference :15-16) ---------------------------------------------------------
GRID_DENSITY = 300
SEARCH_RANGE_MULTIPLIER = 1.5

# --- metres <-> degrees (reference :19-20) --------------------------------------------------
METERS_PER_DEGREE_LAT = 111320.0
METERS_PER_DEGREE_LON = 111320.0


def read_iq_data(filename):
    """Host-side helper kept for API compatibility (reference :26-35): complex64 array of
    (u8 - 127.5)/127.5, or None (with the reference's message) when the file is missing."""
    try:
        raw = np.fromfile(filename, dtype=np.uint8)
    except FileNotFoundError:
        print(f"BŁĄD: Plik '{filename}' nie został znaleziony.")
        return None
    scaled = (raw.astype(np.float32) - 127.5) / 127.5
    return scaled[0::2] + 1j * scaled[1::2]


def find_change_point(amplitude_data, threshold):
    """First index with amplitude > threshold, or None (reference :37-40)."""
    above = np.flatnonzero(np.asarray(amplitude_data) > threshold)
    return above[0] if above.size else None


def meters_to_geographic_degrees(meters_x, meters_y, reference_lat=50.0):
    """(dlat, dlon, dlat_minutes, dlon_minutes) of a metre offset (reference :42-52)."""
    dlat = meters_y / METERS_PER_DEGREE_LAT
    dlon = meters_x / (METERS_PER_DEGREE_LON * math.cos(math.radians(reference_lat)))
    return dlat, dlon, dlat * 60, dlon * 60


def _amplitude_statistics(iq_filename, threshold):
    """('ok', first_index, np.float32 mean amplitude) from the GPU, or ('unreadable' |
    'below_threshold', None, None).  Raises if the HIP library / GPU is unavailable."""
    try:
        # uploaded once per file, shared with the power scan; a file seen for the first time is analysed while it
        # uploads (the amplitude statistics ride on the capture)
        cap = gpsjam.resident_capture(iq_filename, rssi_threshold=float(threshold))
    except FileNotFoundError:
        print(f"BŁĄD: Plik '{iq_filename}' nie został znaleziony.")
        return 'unreadable', None, None
    if cap.nbytes < 2:
        return 'unreadable', None, None
    st = cap.dev.amp_stats(cap, float(threshold))
    if st.first_index < 0:
        return 'below_threshold', None, None
    return 'ok', int(st.first_index), np.float32(st.mean)


def calculate_distance_from_file(iq_filename,
                                 tx_power=DEFAULT_CALIBRATED_TX_POWER,
                                 path_loss_exp=DEFAULT_CALIBRATED_PATH_LOSS_EXPONENT,
                                 frequency_mhz=DEFAULT_SIGNAL_FREQUENCY_MHZ,
                                 threshold=DEFAULT_SIGNAL_THRESHOLD,
                                 verbose=True):
    """Range estimate [m] from the mean received amplitude, or None (reference :54-82)."""
    if verbose:
        print(f"  Analizowanie pliku '{iq_filename}'  ")
    status, _first, avg_amplitude = _amplitude_statistics(iq_filename, threshold)
    if status != 'ok':
        if verbose and status == 'below_threshold':
            print(f"Nie wykryto sygnału z progiem {threshold}.\n")
        return None
    if avg_amplitude == 0:
        return None
    received_power_db = 10 * np.log10(avg_amplitude ** 2)          # float32, as in the reference
    if verbose:
        print(f"Sygnał wykryty. Średnia amplituda: {avg_amplitude:.4f}")
        print(f"Hipotetyczna moc odebrana: {received_power_db:.2f} dB")
    path_loss_at_1m = 20 * np.log10(frequency_mhz) - 27.55
    distance = 10 ** ((tx_power - received_power_db - path_loss_at_1m) / (10 * path_loss_exp))
    if verbose:
        print(f">>> Oszacowana odległość: {distance:.2f} m\n")
    return distance


def perform_grid_search(positions, radii):
    """Grid point minimising sum_k | |p - a_k| - r_k | (reference :88-120); host numpy so
    the argmin (first minimum, row-major) is bit-identical."""
    positions = np.array(positions)
    radii = np.array(radii)
    print(f"Uruchamianie przeszukiwania siatki {GRID_DENSITY}x{GRID_DENSITY}...")
    half_span = np.max(radii) * SEARCH_RANGE_MULTIPLIER
    centre = np.mean(positions, axis=0)
    xs = np.linspace(centre[0] - half_span, centre[0] + half_span, GRID_DENSITY)
    ys = np.linspace(centre[1] - half_span, centre[1] + half_span, GRID_DENSITY)
    gx, gy = np.meshgrid(xs, ys)
    misfit = np.zeros_like(gx)
    for antenna, r in zip(positions, radii):
        misfit += np.abs(np.sqrt((gx - antenna[0]) ** 2 + (gy - antenna[1]) ** 2) - r)
    best = np.unravel_index(np.argmin(misfit), misfit.shape)
    return np.array([gx[best], gy[best]])


def _failure(distances, message, num_antennas):
    return {'success': False, 'distances': distances, 'location_meters': None,
            'location_geographic': None, 'message': message, 'num_antennas': num_antennas}


def triangulate_jammer_location(file_paths,
                                antenna_positions_meters=None,
                                reference_lat=50.00898,
                                reference_lon=19.98287,
                                tx_power=DEFAULT_CALIBRATED_TX_POWER,
                                path_loss_exp=DEFAULT_CALIBRATED_PATH_LOSS_EXPONENT,
                                frequency_mhz=DEFAULT_SIGNAL_FREQUENCY_MHZ,
                                threshold=DEFAULT_SIGNAL_THRESHOLD,
                                verbose=False):
    """Jammer position from per-antenna RSSI ranges (reference :126-229): same arguments,
    same result dict (keys success, distances, location_meters, location_geographic{lat,
    lon, lat_offset_degrees, lon_offset_degrees, lat_offset_minutes, lon_offset_minutes},
    message, num_antennas)."""
    if len(file_paths) < 2:
        return _failure(None, 'Wymagane są co najmniej 2 pliki z danymi anten.', len(file_paths))

    if antenna_positions_meters is None:
        antenna_positions_meters = [np.array([0.0, 0.0]), np.array([0.5, 0.0]),
                                    np.array([0.0, 0.5])][:len(file_paths)]

    distances, used_positions, used_radii = [], [], []
    for index, path in enumerate(file_paths):
        d = calculate_distance_from_file(path, tx_power, path_loss_exp, frequency_mhz, threshold, verbose)
        distances.append(d)
        if d is None:
            continue
        if index < len(antenna_positions_meters):
            used_radii.append(d)
            used_positions.append(np.array(antenna_positions_meters[index]))
        elif verbose:
            print(f"Ostrzeżenie: Brak zdefiniowanej pozycji dla anteny {index}, pomijanie.")

    if len(used_radii) < 2:
        return _failure(distances,
                        'Nie udało się obliczyć poprawnej odległości dla wystarczającej liczby '
                        f'anten (min 2). Sukcesy: {len(used_radii)}', len(file_paths))

    if verbose:
        print(f"Obliczanie lokalizacji metodą Grid Search dla {len(used_positions)} anten.")
        for pos, r in zip(used_positions, used_radii):
            print(f"  Antena [{pos[0]:.1f}, {pos[1]:.1f}] -> r={r:.2f}m")

    best = perform_grid_search(used_positions, used_radii)
    if best is None:
        return _failure(distances, 'Algorytm Grid Search nie zwrócił wyniku.', len(used_radii))

    dlat, dlon, dlat_min, dlon_min = meters_to_geographic_degrees(best[0], best[1], reference_lat)
    return {
        'success': True,
        'distances': distances,
        'location_meters': best.tolist(),
        'location_geographic': {
            'lat': reference_lat + dlat,
            'lon': reference_lon + dlon,
            'lat_offset_degrees': dlat,
            'lon_offset_degrees': dlon,
            'lat_offset_minutes': dlat_min,
            'lon_offset_minutes': dlon_min,
        },
        'message': 'Lokalizacja wyznaczona algorytmem Grid Search (błąd minimalny). '
                   f'x={best[0]:.2f}m, y={best[1]:.2f}m',
        'num_antennas': len(used_radii),
    }
