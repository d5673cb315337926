"""Importable, MI355X-backed counterpart of skrypty/widmo_plot.py of mfkiwl/GPS-JAMMING.

The reference is a script that runs at import with a hard-coded path (widmo_plot.py:7,:96).
Here ``analyze_full_file(filename)`` is a function: it computes what the reference computes
before it starts drawing (widmo_plot.py:12-57,75,85) and returns it:

* ``spectrogram``  -- float32[rows, FFT_SIZE]: 10*log10(Welch PSD + 1e-15) of every 1-s chunk,
  fftshift-ed (scipy.signal.welch(..., nperseg=FFT_SIZE, return_onesided=False), periodic
  Hann, 50 % overlap, per-segment mean removal) -- kernel K2 (``gj_welch_u8``);
* ``mean_spectrum`` -- mean over the rows of the dB spectrogram (:75);
* ``freq_axis_mhz`` -- linspace(-fs/2, fs/2, FFT_SIZE)/1e6 (:74);
* ``histogram``     -- counts of every 100th raw byte per chunk in 256 bins (:35,:85);
* ``duration_sec``.

Plotting (reference :59-93) is left to the caller; ``plot(result)`` draws the same three
panels when matplotlib is available.
"""
import os
import sys

import numpy as np

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # .../gps-jamming_amd
if _PKG_ROOT not in sys.path:
    sys.path.append(_PKG_ROOT)

import gpsjam   # noqa: E402

SAMPLE_RATE = 2.048e6
CHUNK_SIZE = int(SAMPLE_RATE)
FFT_SIZE = 1024


def analyze_full_file(filename, fft_size=FFT_SIZE, sample_rate=SAMPLE_RATE, chunk_size=None):
    chunk_size = int(sample_rate) if chunk_size is None else int(chunk_size)
    file_size = os.path.getsize(filename)
    duration_sec = (file_size // 2) / sample_rate
    print(f"Analiza pliku: {filename}")
    print(f"Rozmiar: {file_size/1024/1024:.2f} MB")
    print(f"Czas trwania: {duration_sec:.2f} sekund")
    print("Przetwarzanie... to może chwilę potrwać.")

    # one upload; K2 and the byte histogram both run on the device-resident capture
    cap = gpsjam.resident_capture(filename, chunk_bytes=0, welch=(chunk_size, fft_size), fs=sample_rate, shift=True,
                                  want_db=True)     # a file seen for the first time: K2 runs while it uploads
    dev = cap.dev
    _, psd_db = dev.welch(cap, chunk_samples=chunk_size, nperseg=fft_size, fs=sample_rate, shift=True, want_db=True)
    histogram = dev.byte_histogram(cap, chunk_size, fft_size, 100)                   # raw_chunk[::100], :35
    return {
        'spectrogram': psd_db,
        'mean_spectrum': psd_db.mean(axis=0) if psd_db.shape[0] else np.zeros(fft_size, np.float32),
        'freq_axis_mhz': np.linspace(-sample_rate / 2, sample_rate / 2, fft_size) / 1e6,
        'histogram': histogram,
        'duration_sec': duration_sec,
    }


def plot(result, filename=""):        # pragma: no cover - needs matplotlib and a display
    import matplotlib.pyplot as plt
    spec = result['spectrogram']
    fig = plt.figure(figsize=(12, 10))
    ax1 = plt.subplot2grid((3, 1), (0, 0), rowspan=2)
    half = SAMPLE_RATE / 2 / 1e6
    img = ax1.imshow(spec, aspect='auto', extent=[-half, half, result['duration_sec'], 0], cmap='inferno')
    ax1.set_title(f'Waterfall (Spektrogram) - {filename}')
    ax1.set_ylabel('Czas [s]')
    ax1.set_xlabel('Częstotliwość względna [MHz]')
    plt.colorbar(img, ax=ax1, label='Moc [dB]')
    ax2 = plt.subplot2grid((3, 2), (2, 0))
    ax2.plot(result['freq_axis_mhz'], result['mean_spectrum'], color='blue')
    ax2.set_title('Średnie Widmo (Cały plik)')
    ax2.grid(True)
    ax3 = plt.subplot2grid((3, 2), (2, 1))
    ax3.bar(np.arange(256), result['histogram'], width=1.0, color='green', alpha=0.7)
    ax3.set_title('Histogram wartości (Raw uint8)')
    ax3.set_xlim(0, 255)
    plt.tight_layout()
    plt.show()


if __name__ == "__main__":            # pragma: no cover
    import sys
    if len(sys.argv) != 2:
        print("usage: python widmo_plot.py <capture.bin>")
        sys.exit(1)
    plot(analyze_full_file(sys.argv[1]), sys.argv[1])
