"""The process-wide cache of resident captures (gpsjam.resident_capture) never holds its lock across an upload:
loads of different files run side by side, a second caller of a file that is being uploaded waits for that upload
instead of starting its own, and a loader whose thread is gone (the GUI stops an analysis with QThread.terminate(),
/root/reference/GpsJammerApp/app/ui_mainwindow.py:818-826) is replaced by the next caller.  Host logic only: the
device is a stand-in whose uploads can be parked."""
import os
import sys
import threading
import time

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "gps-jamming_amd"))

import gpsjam   # noqa: E402


class _Cap:
    def __init__(self, dev, nbytes):
        self.dev, self.nbytes, self.ptr = dev, nbytes, 1

    def free(self):
        self.ptr = 0


class _ParkingDevice:
    """capture() blocks on the gate of the file it is given (when one is installed)."""
    _ctx = object()

    def __init__(self):
        self.gates, self.calls, self.inside = {}, [], threading.Semaphore(0)

    def capture(self, path, *a, **k):
        self.calls.append(os.path.basename(path))
        self.inside.release()
        gate = self.gates.get(os.path.basename(path))
        if gate is not None:
            assert gate.wait(20), "test gate never opened"
        return _Cap(self, os.path.getsize(path))

    ingest = capture


@pytest.fixture
def cache(tmp_path, monkeypatch):
    dev = _ParkingDevice()
    monkeypatch.setattr(gpsjam, "default_device", lambda: dev)
    monkeypatch.setattr(gpsjam, "_resident", {})
    monkeypatch.setattr(gpsjam, "_resident_loading", {})
    paths = []
    for k in range(3):
        p = tmp_path / f"ant{k}.bin"
        p.write_bytes(bytes([128 + k]) * 4096)
        paths.append(str(p))
    return dev, paths


def test_loads_of_different_files_do_not_wait_for_each_other(cache):
    dev, paths = cache
    dev.gates["ant0.bin"] = threading.Event()
    t = threading.Thread(target=gpsjam.resident_capture, args=(paths[0],))
    t.start()
    assert dev.inside.acquire(timeout=10)                 # the first upload is parked inside the device
    t0 = time.perf_counter()
    b = gpsjam.resident_capture(paths[1])                 # a different file: must not queue behind it
    assert time.perf_counter() - t0 < 5 and b.ptr
    assert not gpsjam._resident_lock.locked()
    dev.gates["ant0.bin"].set()
    t.join(10)
    assert not t.is_alive() and len(gpsjam._resident) == 2 and not gpsjam._resident_loading


def test_second_caller_of_a_file_in_flight_shares_the_upload(cache):
    dev, paths = cache
    dev.gates["ant0.bin"] = threading.Event()
    got = []
    ts = [threading.Thread(target=lambda: got.append(gpsjam.resident_capture(paths[0]))) for _ in range(3)]
    ts[0].start()
    assert dev.inside.acquire(timeout=10)
    ts[1].start()
    ts[2].start()
    time.sleep(0.2)                                       # the followers are waiting, not uploading
    assert dev.calls == ["ant0.bin"]
    dev.gates["ant0.bin"].set()
    for t in ts:
        t.join(10)
    assert len(got) == 3 and got[0] is got[1] is got[2] and dev.calls == ["ant0.bin"]


def test_loader_whose_thread_is_gone_is_replaced(cache):
    dev, paths = cache
    st = os.stat(paths[2])
    key = (os.path.realpath(paths[2]), st.st_size, st.st_mtime_ns)
    # what a loader leaves behind when its thread is killed inside the upload: the in-flight entry, never cleared
    dead = threading.Thread(target=lambda: None)
    dead.start()
    tid = dead.native_id
    dead.join()
    for _ in range(200):                                  # the kernel takes the task down a moment after the join returns
        if not gpsjam._thread_alive(tid):
            break
        time.sleep(0.005)
    assert not gpsjam._thread_alive(tid) and gpsjam._thread_alive(threading.get_native_id())
    gpsjam._resident_loading[key] = (threading.Event(), tid)
    t0 = time.perf_counter()
    cap = gpsjam.resident_capture(paths[2])
    assert cap.ptr and time.perf_counter() - t0 < 5 and dev.calls == ["ant2.bin"]
    assert not gpsjam._resident_loading and gpsjam._resident[key] is cap


def test_a_failed_upload_leaves_nothing_in_flight(cache):
    dev, paths = cache

    def boom(*a, **k):
        raise RuntimeError("upload failed")
    dev.capture = boom
    with pytest.raises(RuntimeError):
        gpsjam.resident_capture(paths[0])
    assert not gpsjam._resident_loading and not gpsjam._resident and not gpsjam._resident_lock.locked()
    del dev.capture                                      # the class method again
    assert gpsjam.resident_capture(paths[0]).ptr


def test_release_resident_frees_outside_the_lock(cache):
    dev, paths = cache
    caps = [gpsjam.resident_capture(p) for p in paths]
    gpsjam.release_resident()
    assert all(c.ptr == 0 for c in caps) and not gpsjam._resident



