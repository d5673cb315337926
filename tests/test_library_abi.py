"""CPU-side checks of the C-ABI boundary: the shared library loads, exports every symbol
include/gpsjam.h declares, and its pure size helpers agree with the reference's chunking
rules.  No compute call is made (there is no GPU here)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import gpsjam
from gpsjam import _ffi

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "gpsjam.h")


def header_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gj_[a-z0-9_]+)\s*\(", text)))


def test_library_exists_and_loads():
    assert os.path.exists(_ffi.LIB_PATH), "build with __graft_entry__.build()"
    lib = _ffi.load()
    assert lib.gj_version() == _ffi.GJ_VERSION == 150


def test_every_declared_symbol_is_exported_and_bound():
    syms = header_symbols()
    assert len(syms) >= 30
    out = subprocess.run(["nm", "-D", "--defined-only", _ffi.LIB_PATH], capture_output=True,
                         text=True, check=True).stdout
    exported = set(re.findall(r" T (gj_[a-z0-9_]+)", out))
    missing = [s for s in syms if s not in exported]
    assert not missing, f"declared but not exported: {missing}"
    unbound = [s for s in syms if s not in _ffi.SIGNATURES]
    assert not unbound, f"declared but not bound in _ffi.SIGNATURES: {unbound}"
    extra = [s for s in _ffi.SIGNATURES if s not in syms]
    assert not extra, f"bound but not declared: {extra}"


def test_struct_layouts_match_header():
    assert C.sizeof(_ffi.AmpStats) == 32
    assert C.sizeof(_ffi.Onset) == 32
    assert C.sizeof(_ffi.SynthParams) == 56


def test_status_strings():
    lib = _ffi.load()
    assert lib.gj_strerror(0) == b"ok"
    assert lib.gj_strerror(-4) == b"no such GPU"
    assert lib.gj_strerror(-123) == b"unknown status"


def test_chunk_count_matches_reference_loop():
    lib = _ffi.load()
    # worker.py:216-218 reads until f.read() returns b'' -> ceil division, ragged tail kept
    for nbytes, chunk in [(0, 65536), (1, 65536), (65536, 65536), (65537, 65536),
                          (20 * 65536 + 24691, 65536), (1 << 30, 65536), (40960000, 131072)]:
        assert lib.gj_chunk_count(nbytes, chunk) == -(-nbytes // chunk)
    assert lib.gj_chunk_count(10, 0) == 0


def test_welch_rows_matches_reference_loop():
    lib = _ffi.load()

    def rows_ref(nbytes, chunk_samples, nperseg):        # widmo_plot.py:27-32
        n, off = 0, 0
        while True:
            got = min(2 * chunk_samples, nbytes - off)
            if got < 2 * nperseg:
                break
            n += 1
            off += got
        return n

    for nbytes in (0, 2047, 2048, 8191, 8192, 4096000, 4096000 + 8190, 4096000 + 8192,
                   2 * 4096000, 1 << 30):
        for nperseg in (1024, 4096):
            assert lib.gj_welch_rows(nbytes, 2048000, nperseg) == rows_ref(nbytes, 2048000, nperseg)
    assert lib.gj_welch_rows(1 << 30, 2048000, 4096) == 263


def test_no_gpu_means_loud_failure():
    """Without a GPU the product must fail, not fall back to a CPU path."""
    if gpsjam.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(gpsjam.GpsJamError):
        gpsjam.Device(0)


def test_product_never_imports_oracle():
    pkg = os.path.join(REPO, "gps-jamming_amd")
    bad = []
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(root, f), errors="replace").read()
                if re.search(r"^\s*(from|import)\s+oracle\b|gpsjam_oracle", txt, flags=re.M):
                    bad.append(os.path.join(root, f))
    assert not bad, f"product files reference the oracle: {bad}"


def test_header_is_plain_c_and_cxx(tmp_path):
    """include/gpsjam.h is the boundary: it must compile on its own as C11 (-pedantic) and as C++17, warnings as errors."""
    src = '#include "gpsjam.h"\nint main(void) { return gj_version() == GJ_VERSION ? 0 : 1; }\n'
    for name, cc, std in (("h.c", "gcc", "-std=c11"), ("h.cpp", "g++", "-std=c++17")):
        f = tmp_path / name
        f.write_text(src)
        extra = ["-pedantic"] if cc == "gcc" else []
        r = subprocess.run([cc, std, "-Wall", "-Wextra", "-Werror", *extra, "-I", os.path.join(REPO, "include"), "-fsyntax-only", str(f)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
