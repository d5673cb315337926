"""The library's threaded HOST side under ThreadSanitizer and under AddressSanitizer + UBSan, in the CPU container
(VERDICT r05 "next" 1; SURVEY section 5 "race detection / sanitizers").

csrc/*.hip are compiled host-only (`hipcc --offload-host-only`: every kernel becomes its launch stub) with the sanitizer
and linked with tests/hip_stub -- a stand-in HIP runtime whose streams are worker threads (copies and "kernels" complete
later, in order, on another thread), whose device memory is malloc'ed (so ASan checks every staging copy) and whose
RCCL is an in-process rendezvous.  Kernels do NOT run: this is about locks, lanes, threads, lifetimes -- the host logic
that serves a caller which enters from three threads and kills one (GpsJammerApp/app/worker.py:488-490,610-611,
ui_mainwindow.py:818-826).  Numbers come from the GPU suite only.  Scenarios: tests/hip_stub/san_scenarios.cpp.

GPU-side sanitizers are not available on this pool; nothing here touches a GPU."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
STUB = os.path.join(HERE, "hip_stub")
SCENARIOS = ["threads", "ingest_files", "workspace", "comm", "lanes", "alloc_failures", "api_sweep"]
REPORT_MARKS = ("WARNING: ThreadSanitizer", "ERROR: AddressSanitizer", "ERROR: LeakSanitizer", "runtime error:", "CHECK failed")


@pytest.fixture(scope="module", params=["tsan", "asan"])
def build(request):
    san = request.param
    out = subprocess.run(["make", "-C", STUB, f"SAN={san}", "-j", str(min(8, os.cpu_count() or 1))], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-4000:]
    bdir = os.path.join(STUB, "_build", san)
    exe, rccl = os.path.join(bdir, "san_scenarios"), os.path.join(bdir, "librccl_stub.so")
    assert os.path.exists(exe) and os.path.exists(rccl)
    return san, exe, rccl


@pytest.mark.parametrize("scenario", SCENARIOS)
def test_host_side_is_clean_under_the_sanitizer(build, scenario, tmp_path):
    san, exe, rccl = build
    env = dict(os.environ, GPSJAM_RCCL=rccl, HIP_STUB_DEVICES="2")
    env.pop("GPSJAM_FILL_THREADS", None)
    if san == "tsan":
        env["TSAN_OPTIONS"] = "halt_on_error=0 second_deadlock_stack=1 exitcode=66"
        if scenario == "lanes":
            env["SAN_LANES_PAUSE"] = "1"     # see san_scenarios.cpp: a dead owner never unlocks, TSan cannot see the kernel's hand-over
    else:
        # lanes: a caller that is abandoned inside gj_upload leaves the capture it had allocated and the few bytes of its own
        # stack objects behind -- by design (include/gpsjam.h, "Not covered"); every other scenario runs with leak detection
        env["ASAN_OPTIONS"] = "detect_leaks=%d abort_on_error=0" % (0 if scenario == "lanes" else 1)
        env["UBSAN_OPTIONS"] = "print_stacktrace=1 halt_on_error=1"
    out = subprocess.run([exe, scenario, str(tmp_path)], capture_output=True, text=True, timeout=600, env=env)
    text = out.stdout + out.stderr
    hits = [ln for ln in text.splitlines() if any(m in ln for m in REPORT_MARKS) or ln.startswith("SUMMARY:")]
    assert out.returncode == 0 and not hits, f"{san} {scenario}: rc {out.returncode}\n" + "\n".join(hits[:20]) + "\n...\n" + text[-3000:]
    assert f"{scenario}: ok" in out.stdout
