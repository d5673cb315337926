"""QThread / Signal for the GPSAnalysisThread drop-in.

With PySide6 installed these ARE PySide6.QtCore.QThread / Signal, so the reference GUI
(GpsJammerApp/app/ui_mainwindow.py:684-698) connects its slots exactly as before.  Without
PySide6 (this build image, the GPU box) a small pure-Python stand-in with the same surface is
used: class-level ``Signal(...)`` descriptors with ``connect`` / ``disconnect`` / ``emit``,
and a ``QThread`` with ``start`` / ``run`` / ``isRunning`` / ``wait`` / ``terminate`` / ``quit``
and a ``finished`` signal.
"""
from __future__ import annotations

import threading

try:                                            # pragma: no cover - not installed here
    from PySide6.QtCore import QThread, Signal  # type: ignore
    HAVE_QT = True
except Exception:                               # ImportError or a broken Qt install
    HAVE_QT = False

    class _BoundSignal:
        def __init__(self, arg_types):
            self._types = arg_types
            self._slots = []
            self._lock = threading.Lock()

        def connect(self, slot):
            with self._lock:
                self._slots.append(slot)

        def disconnect(self, slot=None):
            with self._lock:
                if slot is None:
                    self._slots.clear()
                else:
                    self._slots.remove(slot)

        def emit(self, *args):
            if len(args) != len(self._types):
                raise TypeError(f"signal expects {len(self._types)} argument(s), got {len(args)}")
            with self._lock:
                slots = list(self._slots)
            for s in slots:
                s(*args)

    class Signal:                               # noqa: D401 - mirrors the Qt name
        """Class attribute; every instance gets its own bound signal on first access."""

        def __init__(self, *arg_types):
            self._types = arg_types
            self._name = None

        def __set_name__(self, owner, name):
            self._name = "__sig_" + name

        def __get__(self, obj, owner=None):
            if obj is None:
                return self
            bound = obj.__dict__.get(self._name)
            if bound is None:
                bound = obj.__dict__.setdefault(self._name, _BoundSignal(self._types))
            return bound

    class QThread:
        finished = Signal()
        started = Signal()

        def __init__(self, parent=None):
            self._thread = None
            self._parent = parent

        def run(self):                          # overridden by subclasses
            pass

        def _bootstrap(self):
            try:
                self.started.emit()
                self.run()
            finally:
                self.finished.emit()

        def start(self):
            if self.isRunning():
                return
            self._thread = threading.Thread(target=self._bootstrap, daemon=True)
            self._thread.start()

        def isRunning(self) -> bool:
            return self._thread is not None and self._thread.is_alive()

        def isFinished(self) -> bool:
            return self._thread is not None and not self._thread.is_alive()

        def wait(self, msecs: int | None = None) -> bool:
            if self._thread is None:
                return True
            self._thread.join(None if msecs is None else msecs / 1000.0)
            return not self._thread.is_alive()

        def quit(self):
            pass

        def terminate(self):
            # A Python thread cannot be killed; the worker polls stop_requested instead
            # (the GUI sets it right before calling terminate(), ui_mainwindow.py:822-826).
            if hasattr(self, "stop_requested"):
                self.stop_requested = True
