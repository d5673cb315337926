"""Drop-in for GpsJammerApp/app/checkIfJamming.py of mfkiwl/GPS-JAMMING, MI355X-backed.

Same CLI (``<file> <threshold>`` / ``<file> --kalibruj``) and the same output lines -- the
settings dialog parses ``Sugerowany <próg_mocy> ...: <value>`` from stdout
(GpsJammerApp/app/settings_dialog.py:245) -- and the same function names.  The per-chunk
power of the whole file (reference :7-20, chunks of 131 072 bytes, :5) is ONE GPU call
(kernel K1 with the odd-chunk-is-zero rule of :12-13); the event bookkeeping and statistics
are host-side scalar logic.
"""
import os
import sys

import numpy as np

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # .../gps-jamming_amd
if _PKG_ROOT not in sys.path:
    sys.path.append(_PKG_ROOT)

import gpsjam   # noqa: E402

CHUNK_SIZE_BYTES = 131072


def _chunk_powers(raw) -> np.ndarray:
    return gpsjam.default_device().chunk_power(raw, chunk_bytes=CHUNK_SIZE_BYTES, eps=0.0,
                                               odd_chunk_zero=True)


def analyze_chunk_power(raw_uint8_chunk, power_threshold):
    """(is_jamming_now, average_power) of one chunk; odd-sized / empty -> (False, 0.0)."""
    chunk = gpsjam.as_u8(raw_uint8_chunk)
    if chunk.size % 2 != 0 or chunk.size == 0:
        return False, 0.0
    power = gpsjam.default_device().chunk_power(chunk, chunk_bytes=max(chunk.size, 2), eps=0.0,
                                                odd_chunk_zero=True)[0]
    return bool(power > power_threshold), power


def analyze_file_for_jamming(file_path, power_threshold):
    """[(start_sample, end_sample), ...] of the runs whose chunk power exceeds the threshold
    (reference :22-67)."""
    try:
        raw = gpsjam.read_capture(file_path)
        powers = _chunk_powers(raw)
        events, run_start, done = [], None, 0
        for index, power in enumerate(powers):
            nbytes = min(CHUNK_SIZE_BYTES, raw.size - index * CHUNK_SIZE_BYTES)
            new_samples = nbytes // 2
            if new_samples == 0:
                continue
            hot = bool(power > power_threshold) and nbytes % 2 == 0
            if hot and run_start is None:
                run_start = done
            elif not hot and run_start is not None:
                events.append((run_start, done))
                run_start = None
            done += new_samples
        if run_start is not None:
            events.append((run_start, done))
        return events
    except gpsjam.GpsJamLibraryError:
        raise
    except Exception as e:
        print(f"Błąd podczas analizy pliku: {e}")
        return []


def calibrate_file(file_path):
    """Median / max / min of the chunk powers and the suggested threshold median * 4.8
    (reference :69-106)."""
    try:
        powers = _chunk_powers(gpsjam.read_capture(file_path))
        print("--- Kalibracja zakończona ---")
        if powers.size == 0:
            print("Plik jest pusty lub nie zawiera poprawnych danych.")
            return
        floor = np.median(powers)
        suggested = floor * 4.8
        print("\n--- Statystyki mocy (skala cyfrowa I²+Q²) ---")
        print(f"Typowy poziom szumu (Mediana): {floor:.2f}")
        print(f"Moc szczytowa (max):         {np.max(powers):.2f}")
        print(f"Moc minimalna (min):         {np.min(powers):.2f}")
        print(f"\nSugerowany <próg_mocy> (Mediana * 4.8): {suggested:.2f}")
        print(f"Użyj: python {os.path.basename(__file__)} {file_path} {suggested:.2f}")
    except gpsjam.GpsJamLibraryError:
        raise
    except Exception as e:
        print(f"Błąd podczas kalibracji pliku: {e}")


def print_usage_and_exit():
    name = os.path.basename(__file__)
    print("BŁĄD: Niepoprawne użycie.")
    print("\nSposób użycia (Tryb Analizy):")
    print(f"  python {name} <nazwa_pliku.bin> <próg_mocy>")
    print(f"Przykład: python {name} nagranie.iq 120.0")
    print("\nSposób użycia (Tryb Kalibracji):")
    print(f"  python {name} <nazwa_pliku.bin> --kalibruj")
    print(f"Przykład: python {name} nagranie.iq --kalibruj")
    sys.exit(1)


def main(argv):
    if len(argv) != 3:
        print_usage_and_exit()
    path, second = argv[1], argv[2]
    threshold = None
    if second != '--kalibruj':
        try:
            threshold = float(second)
        except ValueError:
            print(f"BŁĄD: <próg_mocy> musi być liczbą (np. '120.0'), a nie '{second}'")
            print_usage_and_exit()
    if not os.path.exists(path):
        print(f"BŁĄD: Nie znaleziono pliku: {path}")
        sys.exit(1)
    if threshold is None:
        print(f"--- Tryb kalibracji: {path} ---")
        print("Proszę czekać, trwa analiza pliku...")
        calibrate_file(path)
        return
    events = analyze_file_for_jamming(path, threshold)
    if not events:
        print("WYNIK: Nie wykryto żadnego jammingu")
        print("jamming_events=[]")
        return
    print(f"WYNIK: Wykryto {len(events)} okres(ów) jammingu:")
    for i, (start, end) in enumerate(events, 1):
        print(f"  Zdarzenie {i}: próbki {start} - {end} (długość: {end - start} próbek)")
    print(f"\njamming_events={events}")


if __name__ == "__main__":
    main(sys.argv)
