"""The C-ABI consumed from plain C (tests/c_abi_smoke.c): gcc against include/gpsjam.h, linked
with the in-tree libgpsjam_hip.so, run on the GPU."""
import os
import subprocess
import tempfile

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_c_program_links_and_runs():
    libdir = os.path.join(REPO, "gps-jamming_amd", "csrc")
    assert os.path.exists(os.path.join(libdir, "libgpsjam_hip.so")), "build the library first (__graft_entry__.build())"
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "c_abi_smoke")
        subprocess.run(["gcc", "-std=c11", "-O2", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"),
                        os.path.join(REPO, "tests", "c_abi_smoke.c"), "-o", exe, "-L", libdir, "-lgpsjam_hip",
                        "-lm", f"-Wl,-rpath,{libdir}"], check=True)
        out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "c_abi_smoke OK" in out.stdout


def test_abandoned_callers_do_not_block_the_context():
    """tests/c_abandoned_caller.c: threads ended by a raw exit inside gj_upload, inside an event wait and as the
    owner of the context mutex; the same context keeps answering correctly and takes their lanes back."""
    libdir = os.path.join(REPO, "gps-jamming_amd", "csrc")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "c_abandoned_caller")
        subprocess.run(["gcc", "-std=gnu11", "-O2", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"),
                        os.path.join(REPO, "tests", "c_abandoned_caller.c"), "-o", exe, "-L", libdir, "-lgpsjam_hip",
                        "-lpthread", f"-Wl,-rpath,{libdir}"], check=True)
        out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "abandoned callers: ok" in out.stdout


def test_c_consumer_of_ingest_and_capture_parts():
    """tests/c_split_ingest.c: gj_ingest_u8 and the capture-part entry points from plain C -- bit-identical to
    upload-then-run and to the unsplit capture."""
    libdir = os.path.join(REPO, "gps-jamming_amd", "csrc")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "c_split_ingest")
        subprocess.run(["gcc", "-std=c11", "-O2", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"),
                        os.path.join(REPO, "tests", "c_split_ingest.c"), "-o", exe, "-L", libdir, "-lgpsjam_hip",
                        f"-Wl,-rpath,{libdir}"], check=True)
        out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "c_split_ingest OK" in out.stdout and "identical to the unsplit capture" in out.stdout
