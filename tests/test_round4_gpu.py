"""Round-4 additions on the GPU: ride-along results keyed by the unpack convention, the worker's scan
serving its own triangulation's amplitude statistics, device identity, live communicator figures."""
import ctypes as C
import io
import os
import sys
from contextlib import redirect_stdout

import numpy as np
import pytest

import gpsjam
from gpsjam.synth import StreamSpec, generate
from oracle import gpsjam_oracle as orc

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(os.path.dirname(HERE), "gps-jamming_amd")
for p in (os.path.join(PKG, "skrypty"), os.path.join(PKG, "GpsJammerApp", "app")):
    if p not in sys.path:
        sys.path.insert(0, p)


# ----------------------------------------------------------------------------- results cache vs set_unpack
def test_ride_along_results_follow_the_unpack_convention(dev):
    """Device.ingest leaves results on the Capture; gj_set_unpack promises to affect every later call on the context.
    After set_unpack(128, 1/128) a call on the ingested capture must NOT hand out numbers computed with 127.5: it
    recomputes, and equals a fresh computation under the new convention.  A cache hit ran no kernel and says so."""
    raw = generate(StreamSpec(seed=91, jam_start=250000, jam_end=1 << 40, jam_sigma=50.0), 400000)
    try:
        with dev.ingest(raw, rssi_threshold=0.0, welch=(100000, 1024)) as cap:
            dev.last_kernel_ms = 123.0
            hits = dev.cache_hits
            old_pm = dev.chunk_power(cap)
            assert dev.cache_hits == hits + 1 and dev.last_kernel_ms == 0.0      # served from the capture: no kernel ran
            old_amp, old_on = dev.amp_stats(cap, 0.0), dev.onset(cap)
            old_psd = dev.welch(cap, chunk_samples=100000, nperseg=1024, want_db=False)[0]
            assert dev.cache_hits == hits + 4
            dev.set_unpack(128.0, 1.0 / 128.0)
            calls = dict(dev.kernel_calls)
            new_pm = dev.chunk_power(cap)
            assert dev.kernel_calls.get("chunk_power", 0) == calls.get("chunk_power", 0) + 1 and dev.last_kernel_ms > 0
            assert not cap.results                                              # everything computed under 127.5 is gone
            new_amp, new_on = dev.amp_stats(cap, 0.0), dev.onset(cap)
            new_psd = dev.welch(cap, chunk_samples=100000, nperseg=1024, want_db=False)[0]
            with dev.capture(raw) as plain:                                     # a fresh computation, nothing cached
                np.testing.assert_array_equal(new_pm, dev.chunk_power(plain))
                fresh_amp, fresh_on = dev.amp_stats(plain, 0.0), dev.onset(plain)
                assert (new_amp.sum, new_amp.first_index, new_amp.count) == (fresh_amp.sum, fresh_amp.first_index, fresh_amp.count)
                assert bytes(new_on) == bytes(fresh_on)
                assert new_psd.tobytes() == dev.welch(plain, chunk_samples=100000, nperseg=1024, want_db=False)[0].tobytes()
            # and the new numbers are the 128-convention's, not the old ones
            i8 = raw.astype(np.float64) - 128.0
            want = (i8[0::2] ** 2 + i8[1::2] ** 2).reshape(-1)
            np.testing.assert_allclose(new_pm[0], want[:32768].mean() + 1e-10, rtol=1e-6)
            assert not np.array_equal(new_pm, old_pm) and new_amp.sum != old_amp.sum
            assert not np.array_equal(new_psd, old_psd)
            assert old_on.start_index > 0 and new_on.start_index > 0
        # back to the default: an ingest under (127.5, 1/127.5) serves again
        dev.set_unpack()
        with dev.ingest(raw, rssi_threshold=0.0) as cap:
            hits = dev.cache_hits
            np.testing.assert_array_equal(dev.chunk_power(cap), old_pm)
            assert dev.cache_hits == hits + 1
    finally:
        dev.set_unpack()


# ----------------------------------------------------------------------------- the worker's ride-along K3
def test_worker_scan_serves_its_own_triangulation(tmp_path, monkeypatch, g3_raws):
    """The worker's power scan ingests the first file with the amplitude threshold ITS OWN triangulation passes
    (0.0, the reference: GpsJammerApp/app/worker.py:598), so triangulate_jammer_location(threshold=0.0) on the same
    files finds the statistics on the captures: three uploads in all, and not one amplitude kernel call."""
    monkeypatch.setattr(gpsjam, "_default", None)
    gpsjam.release_resident()
    import triangulateRSSI
    import worker
    paths = []
    for k, r in enumerate(g3_raws):
        p = tmp_path / f"ant{k}.bin"
        r.tofile(p)
        paths.append(str(p))
    with redirect_stdout(io.StringIO()):
        th = worker.GPSAnalysisThread(paths)
        assert th.TRIANGULATION_RSSI_THRESHOLD == 0.0
        before = gpsjam.Capture.uploads
        th.precalculate_power_profile()
    assert th.power_map_ready and gpsjam.Capture.uploads == before + 1
    dev = gpsjam.default_device()
    cap0 = gpsjam.resident_capture(paths[0])
    assert ("amp_stats", 0.0) in cap0.results            # rides on the capture since the scan
    calls, hits = dict(dev.kernel_calls), dev.cache_hits
    with redirect_stdout(io.StringIO()):
        res = triangulateRSSI.triangulate_jammer_location(
            file_paths=th.get_test_files_for_triangulation(),
            antenna_positions_meters=[np.array(th.antenna_positions[k]) for k in ("antenna1", "antenna2", "antenna3")],
            threshold=th.TRIANGULATION_RSSI_THRESHOLD, verbose=False)
    assert res["success"] and res["num_antennas"] == 3
    assert gpsjam.Capture.uploads == before + 3          # files 1 and 2 came in (analysed while they uploaded); file 0 did not
    assert dev.kernel_calls.get("amp_stats", 0) == calls.get("amp_stats", 0)     # no amplitude kernel call at all
    assert dev.cache_hits == hits + 3
    want = orc.triangulate(g3_raws, threshold=0.0)
    np.testing.assert_allclose(res["distances"], want["distances"], rtol=1e-5)
    assert res["location_meters"] == want["location_meters"]
    # the CLI's default threshold (0.1) is a different question: it is computed, on the resident captures
    with redirect_stdout(io.StringIO()):
        res2 = triangulateRSSI.triangulate_jammer_location(paths, threshold=0.1)
    assert res2["success"] and gpsjam.Capture.uploads == before + 3
    assert dev.kernel_calls.get("amp_stats", 0) == calls.get("amp_stats", 0) + 3
    gpsjam.release_resident()


# ----------------------------------------------------------------------------- device identity, live communicator
def test_device_identity_names_the_physical_gpu(dev):
    ident = dev.identity()
    fields = dict(tok.split("=", 1) for tok in ident.split())
    assert set(fields) == {"pci", "uuid", "hip"}
    dom, bus, rest = fields["pci"].split(":")
    assert len(dom) == 4 and len(bus) == 2 and "." in rest and fields["hip"] == "0"
    with gpsjam.Device(0) as other:
        assert other.identity() == ident                # two contexts on one GPU: one identity


def test_communicator_reports_live_figures(dev):
    """gj_comm_rank / gj_comm_device read ncclCommUserRank / ncclCommCount / ncclCommCuDevice of the live communicator;
    after the context is gone the handle answers -1 of 0."""
    from gpsjam.comm import Communicator
    d2 = gpsjam.Device(0)
    comm = Communicator(d2, 0, 1, port=29731)
    assert comm.live() == (0, 1, 0)
    a, b = d2.alloc(256), d2.alloc(256)
    a.upload(np.arange(256, dtype=np.uint8))
    comm.allgather(a, 256, b)
    d2.synchronize()
    assert b.download(np.uint8, 256).tolist() == list(range(256))
    h = comm._h
    a.free()
    b.free()
    d2.close()                                            # takes the communicator down, the handle stays valid
    r, n = C.c_int(7), C.c_int(7)
    assert dev._lib.gj_comm_rank(h, C.byref(r), C.byref(n)) == 0 and (r.value, n.value) == (-1, 0)
    comm.dev = dev                                        # close() only frees the handle now
    comm.close()


# ----------------------------------------------------------------------------- streams on hardware queues of their own
def test_stream_beside_finds_a_stream_that_overlaps(dev):
    """gpsjam.streams: a stream is tested against another by keeping that one busy with a spinning wave
    (gj_probe_busy_dev) and recording an event on the candidate.  A stream never runs beside itself; stream_beside
    returns one that runs beside the main stream, and beside two streams at once."""
    import torch
    from gpsjam import streams
    main = torch.cuda.Stream()
    dev.set_stream(main.cuda_stream)
    try:
        assert streams.runs_beside(dev, main, main) is False          # the same queue by definition
        side = streams.stream_beside([(dev, main)])
        assert streams.runs_beside(dev, main, side) is True
        with gpsjam.Device(0) as dev2:
            dev2.set_stream(side.cuda_stream)
            third = streams.stream_beside([(dev, main), (dev2, side)])
            assert streams.runs_beside(dev, main, third) and streams.runs_beside(dev2, side, third)
        # the busy kernel holds the stream for about the time asked, and nothing else
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(main)
        dev.probe_busy_dev(3.0)
        b.record(main)
        torch.cuda.synchronize()
        assert 2.5 < a.elapsed_time(b) < 6.0
        with pytest.raises(gpsjam.GpsJamError):
            dev.probe_busy_dev(1000.0)
    finally:
        dev.set_stream(None, external=False)


# ----------------------------------------------------------------------------- the ingest follows a stream switch
def test_ingest_follows_a_stream_switch_made_mid_call(dev):
    """Another thread may call gj_set_stream while gj_ingest_* is between two of its lock sections (AntennaStream and
    SplitStreams do at construction).  Emulated exactly at the worst place: the wait hook at the head of the staged copy --
    after the ingest has read the context's stream and queued its first work on it, before a single piece has landed --
    points the context at ANOTHER stream.  The kernels the ingest launches from then on run on the new stream; they
    must still wait for the pieces (and for what was queued on the old stream): same results as an undisturbed ingest."""
    import torch
    n = 24 << 20                                            # samples: 48 MiB, a dozen pieces
    raw = generate(StreamSpec(seed=93, jam_start=9_000_000, jam_end=1 << 40, jam_sigma=55.0), n)
    with dev.ingest(raw, rssi_threshold=0.0, welch=(2048000, 1024)) as cap:
        want = (dev.chunk_power(cap).copy(), dev.welch(cap, nperseg=1024, want_db=False)[0].copy(), dev.amp_stats(cap, 0.0),
                dev.onset(cap))
    other = torch.cuda.Stream()
    switched = []

    @C.CFUNCTYPE(None, C.c_void_p, C.c_int)
    def hook(_arg, site):
        if site == 3 and not switched:
            switched.append(True)
            dev.set_stream(other.cuda_stream)               # "another thread": gj_set_stream between two lock sections

    dev._check(dev._lib.gj_debug_set_wait_hook(dev._ctx, C.cast(hook, C.c_void_p), None))
    try:
        for _ in range(3):                                  # a race that is lost shows up as garbage in early chunks
            switched.clear()
            dev.set_stream(None, external=False)
            with dev.ingest(raw, rssi_threshold=0.0, welch=(2048000, 1024)) as cap:
                assert switched, "the hook never fired: the ingest did not take the staged path"
                np.testing.assert_array_equal(dev.chunk_power(cap), want[0])
                assert dev.welch(cap, nperseg=1024, want_db=False)[0].tobytes() == want[1].tobytes()
                got_amp, got_on = dev.amp_stats(cap, 0.0), dev.onset(cap)
                assert (got_amp.sum, got_amp.count, got_amp.first_index) == (want[2].sum, want[2].count, want[2].first_index)
                assert bytes(got_on) == bytes(want[3])
    finally:
        dev._check(dev._lib.gj_debug_set_wait_hook(dev._ctx, None, None))
        dev.set_stream(None, external=False)
