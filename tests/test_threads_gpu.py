"""The library may be entered from several host threads (the GUI's QThread, its
HTTP handler thread and the triangulation thread): calls on ONE context each work in a lane of
their own (the context's mutex is held only while kernels are enqueued), different contexts are
independent.  Both patterns, concurrently, against the oracle."""
import threading

import numpy as np
import pytest

import gpsjam
from gpsjam.synth import StreamSpec, generate
from oracle import gpsjam_oracle as orc

pytestmark = pytest.mark.gpu


def test_concurrent_host_threads(dev):
    raws = [generate(StreamSpec(seed=40 + k, jam_start=120000, jam_end=1 << 40, jam_sigma=40.0 + 5 * k), 300000 + 1111 * k)
            for k in range(4)]
    want_pm = [orc.chunk_power(r) for r in raws]
    want_amp = [orc.rssi_amp_stats(r, 0.0)[1] for r in raws]
    own = [gpsjam.Device(0) for _ in range(2)]
    errors = []

    def work(k, device):
        try:
            for _ in range(6):
                pm = device.chunk_power(raws[k])
                np.testing.assert_allclose(pm, want_pm[k], rtol=1e-6)
                st = device.amp_stats(raws[k], 0.0)
                np.testing.assert_allclose(st.mean, want_amp[k], rtol=1e-6)
                psd, _ = device.welch(raws[k], chunk_samples=100000, nperseg=1024, want_db=False)
                assert np.isfinite(psd).all()
        except Exception as e:   # noqa: BLE001 -- reported below, on the main thread
            errors.append((k, repr(e)))

    # threads 0 and 1 share the fixture's context, 2 and 3 have one each
    threads = [threading.Thread(target=work, args=(0, dev)), threading.Thread(target=work, args=(1, dev)),
               threading.Thread(target=work, args=(2, own[0])), threading.Thread(target=work, args=(3, own[1]))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for d in own:
        d.close()
    assert not errors, errors
