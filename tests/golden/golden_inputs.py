"""Seeded inputs of the golden vectors (regenerated, never committed as .bin).

``make_golden.py`` feeds these to the reference; the tests feed the very same bytes to
the oracle and to the HIP path and compare with the stored outputs.  The sha256 of
every stream is stored in golden_meta.json so a drift of the generator is caught.
"""
import numpy as np

from gpsjam.synth import StreamSpec, generate

G3_POSITIONS = [[0.0, 0.0], [0.5, 0.0], [0.0, 0.5]]
G4_DELAYS = (0, 3, -5)
G4_SLICES = (50000, 1 << 19)
G4_ONSET = 300000


def g1_stream() -> np.ndarray:
    """20 full 65 536-byte chunks + a ragged, odd-length tail; 4-chunk burst."""
    nbytes = 20 * 65536 + 24691
    spec = StreamSpec(seed=101, antenna=0, jam_start=10 * 32768, jam_end=14 * 32768,
                      noise_sigma=8.0, jam_sigma=40.0)
    return generate(spec, (nbytes + 1) // 2)[:nbytes]


def g2_stream() -> np.ndarray:
    """One full 1-s chunk + a 300 000-sample partial chunk, DC offset, burst in the
    middle of the first chunk."""
    spec = StreamSpec(seed=202, antenna=0, jam_start=700000, jam_end=1500000,
                      noise_sigma=6.25, jam_sigma=25.0, dc_i_q8=3 * 256 + 77,
                      dc_q_q8=-2 * 256 - 30)
    return generate(spec, 2048000 + 300000)


def g3_streams():
    """Three antennas, same burst at different strengths, starting at sample 50 000."""
    out = []
    for k, sig in enumerate((60.0, 42.0, 30.0)):
        spec = StreamSpec(seed=303, antenna=k, delay=0, jam_start=50000, jam_end=1 << 40,
                          noise_sigma=6.25, jam_sigma=sig)
        out.append(generate(spec, 150000))
    return out


def g4_streams():
    """Three antennas; common broadband source switched on at G4_ONSET and delayed by
    G4_DELAYS samples; long enough for 2^19-sample slices."""
    n = G4_ONSET + (1 << 19) + 4000
    out = []
    for k, d in enumerate(G4_DELAYS):
        spec = StreamSpec(seed=404, antenna=k, delay=d, jam_start=G4_ONSET, jam_end=1 << 40,
                          noise_sigma=6.25, jam_sigma=60.0)
        out.append(generate(spec, n))
    return out


def g5_scenario():
    """Power-scan state the detector replay starts from (set directly on the thread object:
    the telemetry's byte counter runs to 169 MB, no capture of that size is shipped)."""
    total = 170_000_000
    n = -(-total // 65536)
    idx = np.arange(n)
    pm = (70.0 + 3.0 * np.sin(idx * 0.37)).astype(np.float32)
    pm[1000:1400] = 900.0 + (idx[1000:1400] % 7).astype(np.float32)
    return {"total_file_bytes": total, "power_map": pm,
            "baseline": np.percentile(pm, 5),
            "ranges": [(np.int64(1000 * 65536), np.int64(1400 * 65536))]}


def g5_records(reduced_log):
    """The recorded telemetry followed by a deterministic synthetic stretch: good fixes with
    C/N0 around 45 dB-Hz, a 6-s C/N0 collapse (quality event, confirmed after 2.5 s), recovery
    (closed after 2 s clean), an altitude excursion, and records without a position block."""
    recs = list(reduced_log)
    t0 = float(recs[-1]["elapsed_time"])
    buff = int(recs[-1]["position"]["buffcnt"])
    out = []
    for i in range(420):
        t = t0 + 0.1 * (i + 1)
        buff += 327680
        snr_level = 45.0 + ((i * 7) % 5) * 0.5
        if 150 <= i < 210:
            snr_level -= 20.0
        hgt = 230.0 if not (300 <= i < 340) else 25000.0
        obs = [{"snr": snr_level + k * 0.25, "residual": float((i + k) % 9) * (120.0 if 250 <= i < 255 else 1.0)}
               for k in range(5)]
        rec = {"elapsed_time": round(t, 3),
               "position": {"nsat": 5, "lat": 50.0172 + i * 1e-6, "lon": 19.9401 + i * 1e-6, "hgt": hgt,
                            "gdop": 2.5, "clk_bias": 1e-4, "buffcnt": buff},
               "observations": obs}
        if i % 97 == 13:
            rec = {"elapsed_time": round(t, 3), "observations": []}          # no position block
        if i % 101 == 50:
            rec["elapsed_time"] = "not-a-number"                              # float() failure is swallowed
        out.append(rec)
    return recs + out
