"""Reproducible synthetic RTL-SDR captures (uint8 interleaved I/Q), integer-only.

The value model follows the reference's simulator chain
(simulate/frontend/weaken_gps.py:4-5,27-28 and add_jammer_and_mix.py:9-12,170-177):
a Gaussian receiver floor of a few LSB, plus a broadband jammer burst that is COMMON
to all antennas and reaches antenna ``a`` with an integer delay (so that TDOA lags
are known), truncated toward zero, clipped to int8 and offset by +128.

Every operation is exact integer arithmetic on a counter-based hash (splitmix64), so
the HIP generator in ``csrc/synth.hip`` produces bit-identical bytes on the GPU box
without shipping gigabytes of fixtures.  The approximate Gaussian is the centred sum
of eight uniform 16-bit lanes (Irwin-Hall, sigma = 53509.9).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

_M64 = (1 << 64) - 1
_GAMMA = 0x9E3779B97F4A7C15
_MUL1 = 0xBF58476D1CE4E5B9
_MUL2 = 0x94D049BB133111EB
_KEY_NOISE = 0xA5A5A5A55A5A5A5A
_KEY_COMMON = 0xC0FFEE0DDF00D5EE
IRWIN_HALL_SIGMA = 53509.94          # sqrt(8 * (65536**2 - 1) / 12)


def _sm64_int(x: int) -> int:
    x = (x + _GAMMA) & _M64
    z = x
    z = ((z ^ (z >> 30)) * _MUL1) & _M64
    z = ((z ^ (z >> 27)) * _MUL2) & _M64
    return z ^ (z >> 31)


def _sm64(x: np.ndarray) -> np.ndarray:
    x = x + np.uint64(_GAMMA)
    z = (x ^ (x >> np.uint64(30))) * np.uint64(_MUL1)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(_MUL2)
    return z ^ (z >> np.uint64(31))


def _lanes_sum(h: np.ndarray) -> np.ndarray:
    m = np.uint64(0xFFFF)
    return ((h & m) + ((h >> np.uint64(16)) & m) + ((h >> np.uint64(32)) & m)
            + (h >> np.uint64(48))).astype(np.int64)


def _gauss(key: int, idx: np.ndarray) -> np.ndarray:
    """Centred Irwin-Hall(8) variate for every counter in ``idx`` (int64, may be
    negative: counters wrap modulo 2^64 exactly as on the GPU)."""
    c = idx.astype(np.int64).view(np.uint64) * np.uint64(2) + np.uint64(key)
    return _lanes_sum(_sm64(c)) + _lanes_sum(_sm64(c + np.uint64(1))) - 262140


def gain_k(sigma_lsb: float) -> int:
    """Fixed-point multiplier so that (G * k) >> 16 is a q8 (1/256 LSB) amplitude of
    standard deviation ``sigma_lsb``."""
    return int(round(sigma_lsb * 256.0 * 65536.0 / IRWIN_HALL_SIGMA))


@dataclass
class StreamSpec:
    """One antenna capture.  ``jam_start``/``jam_end`` are in SOURCE time (samples);
    antenna sees the source delayed by ``delay`` samples."""
    seed: int = 1234
    antenna: int = 0
    delay: int = 0
    jam_start: int = 0
    jam_end: int = 0
    noise_sigma: float = 6.25
    jam_sigma: float = 40.0
    dc_i_q8: int = 0
    dc_q_q8: int = 0

    @property
    def noise_k(self) -> int:
        return gain_k(self.noise_sigma)

    @property
    def jam_k(self) -> int:
        return gain_k(self.jam_sigma)

    @property
    def key_noise(self) -> int:
        return _sm64_int((self.seed ^ _KEY_NOISE) + self.antenna & _M64)

    @property
    def key_common(self) -> int:
        return _sm64_int(self.seed ^ _KEY_COMMON)


def generate(spec: StreamSpec, n_samples: int, first_sample: int = 0,
             block: int = 1 << 20) -> np.ndarray:
    """uint8[2*n_samples] interleaved I/Q for samples [first_sample, first_sample+n)."""
    out = np.empty(2 * n_samples, np.uint8)
    kn, kj = spec.noise_k, spec.jam_k
    key_n, key_c = spec.key_noise, spec.key_common
    dc = np.array([spec.dc_i_q8, spec.dc_q_q8], np.int64)
    for lo in range(0, n_samples, block):
        hi = min(n_samples, lo + block)
        n = np.arange(first_sample + lo, first_sample + hi, dtype=np.int64)
        comp = np.tile(np.array([0, 1], np.int64), hi - lo)
        nn = np.repeat(n, 2)
        v = (_gauss(key_n, 2 * nn + comp) * kn) >> 16
        src = nn - spec.delay
        active = (src >= spec.jam_start) & (src < spec.jam_end)
        if active.any():
            j = (_gauss(key_c, 2 * src + comp) * kj) >> 16
            v = v + np.where(active, j, 0)
        v = v + dc[comp]
        t = np.where(v >= 0, v >> 8, -((-v) >> 8))
        out[2 * lo:2 * hi] = (np.clip(t, -128, 127) + 128).astype(np.uint8)
    return out
