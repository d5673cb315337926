"""RCCL collectives of the sharded path without torch.distributed (gj_comm_* of the C-ABI).

One ``Communicator`` per process / GPU.  The 128-byte RCCL unique id is made on rank 0 and
handed to the other ranks over one TCP connection each (MASTER_ADDR / MASTER_PORT of the
launcher's environment, the same rendezvous variables torch.distributed.run exports); after
that every byte moves over xGMI.  ``gather`` / ``bcast`` enqueue on the Device's current HIP
stream and return at once.
"""
from __future__ import annotations

import ctypes as C
import os
import socket
import time
from typing import Optional

from . import _ffi, _ptr

ID_BYTES = _ffi.GJ_COMM_ID_BYTES
_PORT_OFFSET = 17          # next to, not on, the port torch's own store may be using


def _serve_id(uid: bytes, addr: str, port: int, n_clients: int, timeout: float):
    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    srv.bind((addr, port))
    srv.listen(n_clients)
    srv.settimeout(timeout)
    try:
        for _ in range(n_clients):
            conn, _ = srv.accept()
            with conn:
                conn.sendall(uid)
    finally:
        srv.close()


def _fetch_id(addr: str, port: int, timeout: float) -> bytes:
    deadline = time.monotonic() + timeout
    while True:
        try:
            with socket.create_connection((addr, port), timeout=5.0) as s:
                buf = b""
                while len(buf) < ID_BYTES:
                    part = s.recv(ID_BYTES - len(buf))
                    if not part:
                        raise ConnectionError("rank 0 closed the id connection early")
                    buf += part
                return buf
        except (ConnectionRefusedError, socket.timeout, ConnectionError):
            if time.monotonic() > deadline:
                raise
            time.sleep(0.05)


class Communicator:
    def __init__(self, dev, rank: Optional[int] = None, world_size: Optional[int] = None,
                 addr: Optional[str] = None, port: Optional[int] = None, timeout: float = 120.0):
        self.dev = dev
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world_size is None else int(world_size)
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(port if port is not None else int(os.environ.get("MASTER_PORT", "29500")) + _PORT_OFFSET)
        lib = dev._lib
        if addr in ("127.0.0.1", "localhost"):        # one node: see bench.py
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
            os.environ.setdefault("NCCL_IB_DISABLE", "1")
        uid = C.create_string_buffer(ID_BYTES)
        if self.rank == 0:
            rc = lib.gj_comm_unique_id(uid)
            if rc != 0:
                raise _ffi.GpsJamError(rc, "gj_comm_unique_id: " + lib.gj_strerror(rc).decode()
                                       + " (librccl could not be loaded?)")
            if self.world > 1:
                _serve_id(uid.raw, addr, port, self.world - 1, timeout)
        else:
            uid = C.create_string_buffer(_fetch_id(addr, port, timeout), ID_BYTES)
        h = C.c_void_p()
        dev._check(lib.gj_comm_init_rank(dev._ctx, uid, self.rank, self.world, C.byref(h)))
        self._h = h

    def live(self):
        """(rank, ranks, HIP device) as the live RCCL communicator reports them (ncclCommUserRank / ncclCommCount /
        ncclCommCuDevice) -- not what this object was constructed with."""
        r, n, d = C.c_int(-1), C.c_int(0), C.c_int(-1)
        self.dev._check(self.dev._lib.gj_comm_rank(self._h, C.byref(r), C.byref(n)))
        self.dev._check(self.dev._lib.gj_comm_device(self._h, C.byref(d)))
        return r.value, n.value, d.value

    def gather(self, d_send, nbytes: int, d_recv=None, root: int = 0):
        """Every rank sends nbytes; ``root`` receives world * nbytes in rank order."""
        self.dev._check(self.dev._lib.gj_comm_gather_dev(self._h, _ptr(d_send), int(nbytes), _ptr(d_recv) or None, root))

    def allgather(self, d_send, nbytes: int, d_recv):
        """Every rank sends nbytes and receives world * nbytes in rank order."""
        self.dev._check(self.dev._lib.gj_comm_allgather_dev(self._h, _ptr(d_send), int(nbytes), _ptr(d_recv)))

    def bcast(self, d_buf, nbytes: int, root: int = 0):
        self.dev._check(self.dev._lib.gj_comm_bcast_dev(self._h, _ptr(d_buf), int(nbytes), root))

    def close(self):
        if getattr(self, "_h", None):
            self.dev._lib.gj_comm_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
