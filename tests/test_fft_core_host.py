"""Runs csrc/fft_core.h -- the very header the gfx950 kernels are built from -- on the CPU,
one emulated thread at a time (tests/host_fft_emul.cpp), for every supported transform size
against an O(N^2) double-precision DFT.  Catches index-map / twiddle / butterfly mistakes
before a GPU minute is spent."""
import os
import subprocess
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_block_fft_schedule_on_host():
    src = os.path.join(REPO, "tests", "host_fft_emul.cpp")
    inc = os.path.join(REPO, "gps-jamming_amd", "csrc")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "fft_emul")
        subprocess.run(["g++", "-O2", "-std=c++17", "-I", inc, src, "-o", exe], check=True)
        out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("OK")
    assert out.stdout.count("rel_err") == 9
