"""Integer-exact restatements (numpy, CPU) of what K1, K3 and K4 PROMISE -- test infrastructure, never imported by the
product.  The oracle (oracle/gpsjam_oracle.py) restates the reference's float arithmetic; these restate the library's
contract in include/gpsjam.h: sums of (2I-255)^2 + (2Q-255)^2 are exact integers, rounded once.  The GPU tests compare
the kernels with BOTH: the oracle says "the reference's answer", these say "the bits the header documents", so a test
that compares two entry points of the library with each other never stands alone (VERDICT r05 "next" 3).
tests/test_exact_restatement.py (CPU) pins these against the oracle and the golden vectors."""
import numpy as np

GUARD = 1e-6      # kOnsetGuard (include/gpsjam.h, gj_onset.guard_index)


def msq(raw: np.ndarray) -> np.ndarray:
    """4 |z|^2 = (2I-255)^2 + (2Q-255)^2 per I/Q pair, int64; a trailing odd byte is not used."""
    n = raw.size // 2
    v = 2 * raw[:2 * n].astype(np.int64) - 255
    return v[0::2] ** 2 + v[1::2] ** 2


def chunk_power(raw: np.ndarray, chunk_bytes: int, eps=1e-10, odd_chunk_zero=False) -> np.ndarray:
    """K1: float32(sum / (4 n)) + float32(eps) per chunk; NaN for a chunk without a pair (numpy's mean of an empty
    slice, worker.py:228-235); with odd_chunk_zero an odd-length chunk is 0 (checkIfJamming.py:52-55)."""
    out = []
    for off in range(0, raw.size, chunk_bytes):
        piece = raw[off:off + chunk_bytes]
        n = piece.size // 2
        if odd_chunk_zero and (piece.size & 1):
            out.append(np.float32(0.0))
        elif n == 0:
            out.append(np.float32(np.nan))
        else:
            out.append(np.float32(np.float32(int(msq(piece).sum()) / (4.0 * n)) + np.float32(eps)))
    return np.array(out, np.float32)


def onset(raw: np.ndarray, noise_samples=200000, window=1000, factor=50.0) -> dict:
    """K4 (gj_onset): exact integer window sums against float32(noise) * float32(factor).
    Returns start / guard (+ window // 2, -1 = none), noise, thr (float32) and margin_hit (float32)."""
    m = msq(raw)
    n = m.size
    none = dict(start=-1, guard=-1, noise=np.float32(0), thr=np.float32(0), hit=np.float32(0))
    if n < noise_samples + window:
        return none
    noise = np.float32(int(m[:noise_samples].sum()) / (4.0 * noise_samples))
    if noise == 0:
        noise = np.float32(1e-9)
    thr_f = np.float32(noise * np.float32(factor))
    thr = float(thr_f)
    c = np.concatenate([[0], np.cumsum(m)])
    S = c[window:] - c[:-window]                       # n - window + 1 window sums, exact
    ma = S.astype(np.float64) * (0.25 / window)
    hits = np.flatnonzero(ma > thr)
    band = np.flatnonzero(ma > thr * (1.0 - GUARD))
    out = dict(none, noise=noise, thr=thr_f)
    if hits.size:
        i0 = int(hits[0])
        out["start"] = i0 + window // 2
        out["hit"] = np.float32((float(S[i0]) * (0.25 / window) - thr) / thr)
    if band.size:
        out["guard"] = int(band[0]) + window // 2
    return out


def amp_stats(raw: np.ndarray, threshold: float) -> dict:
    """K3 (gj_amp_stats): amplitude = float32(sqrt(float32(m))) * float32(1/255); first index with amplitude >
    threshold (float32 compare), count from there on, and the sum from there on in float64 (the kernels add float32
    partial sums of eight samples: compare at 2e-7)."""
    m = msq(raw)
    hs = np.float32((1.0 / 127.5) * 0.5)
    r = np.sqrt(m.astype(np.float32))
    hit = np.flatnonzero((r * hs) > np.float32(threshold))
    if hit.size == 0:
        return dict(first=-1, count=0, sum=0.0, mean=np.float32(0))
    k = int(hit[0])
    total = float(r[k:].astype(np.float64).sum()) * float(hs)
    return dict(first=k, count=m.size - k, sum=total, mean=np.float32(total / (m.size - k)))
