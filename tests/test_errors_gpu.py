"""Error behaviour of the C-ABI on a live context: bad parameters come back as status codes with
a message (surfaced as GpsJamError by the ctypes layer), never as a crash, and the context stays
usable afterwards."""
import numpy as np
import pytest

import gpsjam
from gpsjam.synth import StreamSpec, generate

pytestmark = pytest.mark.gpu

GJ_ERR_INVALID, GJ_ERR_UNSUPPORTED = -1, -5


def test_bad_parameters_are_reported_and_context_survives(dev):
    raw = generate(StreamSpec(seed=3), 50000)
    with pytest.raises(gpsjam.GpsJamError) as e:
        dev.welch(raw, chunk_samples=20000, nperseg=1000)            # not a power of two
    assert e.value.status == GJ_ERR_UNSUPPORTED and "nperseg" in str(e.value)
    with pytest.raises(gpsjam.GpsJamError) as e:
        dev.welch(raw, chunk_samples=20000, nperseg=8192)            # above the supported size
    assert e.value.status == GJ_ERR_UNSUPPORTED
    with pytest.raises(gpsjam.GpsJamError) as e:
        dev.onset(raw, 1000, 9000, 50.0)                             # window > 8192
    assert e.value.status == GJ_ERR_UNSUPPORTED and "window" in str(e.value)
    with pytest.raises(gpsjam.GpsJamError) as e:
        dev.onset(raw, 0, 1000, 50.0)
    assert e.value.status == GJ_ERR_INVALID
    with pytest.raises(gpsjam.GpsJamError) as e:
        dev.chunk_power(raw, chunk_bytes=0)
    assert e.value.status == GJ_ERR_INVALID
    with pytest.raises(gpsjam.GpsJamError) as e:
        dev.xcorr_lags([raw] * 17, [(0, 1)])                         # more than GJ_MAX_ANTENNAS
    assert e.value.status == GJ_ERR_INVALID
    d_psd = dev.alloc(4 * 4096)
    with pytest.raises(gpsjam.GpsJamError) as e:                     # chunk shorter than one segment
        dev.welch_dev(dev.alloc(raw.size).upload(raw), raw.size, 1000, 4096, 2.048e6, d_psd)
    assert e.value.status == GJ_ERR_UNSUPPORTED
    with pytest.raises(gpsjam.GpsJamError) as e:                     # one chunk = 2^31 samples: 32-bit chunk offsets
        dev.welch_dev(dev.alloc(raw.size).upload(raw), raw.size, 1 << 31, 4096, 2.048e6, d_psd)
    assert e.value.status == GJ_ERR_UNSUPPORTED
    # and the context still works
    pm = dev.chunk_power(raw)
    assert pm.shape == (2,) and np.isfinite(pm).all()


def test_empty_and_tiny_inputs(dev):
    assert dev.chunk_power(np.zeros(0, np.uint8)).size == 0
    psd, _ = dev.welch(np.zeros(10, np.uint8), nperseg=1024)
    assert psd.shape == (0, 1024)
    st = dev.amp_stats(np.zeros(0, np.uint8), 0.0)
    assert st.first_index == -1 and st.count == 0
    assert dev.onset(np.zeros(100, np.uint8), 200000, 1000, 50.0).start_index == -1
