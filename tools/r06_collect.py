#!/usr/bin/env python3
"""Copy the judged evidence of a `tools/r06_profile.sh <tag>` run from gpurun_out/<tag>/ into profiles/ (tracked):
    python tools/r06_collect.py r06p"""
import glob
import os
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06p"
src = os.path.join(REPO, "gpurun_out", tag)
dst = os.path.join(REPO, "profiles")
commit = subprocess.run(["git", "-C", REPO, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()


def cp(a, b):
    a = os.path.join(src, a)
    if os.path.exists(a) and os.path.getsize(a):
        shutil.copyfile(a, os.path.join(dst, b))
        print("  ", b)
    else:
        print("  MISSING", a)


for a, b in (("bench.json", "r06_bench.json"), ("force_exchange.json", "r06_force_exchange_n1.json"),
             ("weak_emulated8.json", "r06_weak_emulated8.json"), ("split_n1.json", "r06_split_n1.json"),
             ("split_emulated8_rank0.json", "r06_split_emulated8_rank0.json"), ("split_emulated4.json", "r06_split_emulated4_rank0.json"),
             ("split_emulated8_rank1.json", "r06_split_emulated8_rank1.json"), ("split_emulated8_rank2.json", "r06_split_emulated8_rank2.json"),
             ("split_emulated8_rank7.json", "r06_split_emulated8_rank7.json"), ("deployment.txt", "r06_deployment_probe.txt"),
             ("k2_ab_probe.txt", "r06_k2_ab_probe.txt")):
    cp(a, b)


def stats(sub, name, pick=0):
    f = sorted(glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True))
    if len(f) > pick:
        shutil.copyfile(f[pick], os.path.join(dst, name))
        print("  ", name)
    else:
        print("  MISSING kernel stats under", sub)


stats("bench_trace", "r06_bench_kernel_stats.csv")
stats("split8_trace", "r06_split_emulated8_kernel_stats.csv")
stats("cscan_big", "r06_solo_cscan_gib_kernel_stats.csv")
stats("cscan_small", "r06_solo_cscan_10s_kernel_stats.csv")
stats("solo_xcorr3", "r06_solo_xcorr3_kernel_stats.csv")
stats("solo_k4", "r06_solo_onset_alone_gib_kernel_stats.csv")
stats("dep_trace", "r06_deployment_kernel_stats.csv")
# K2 solo at 4096 and 1024: two runs each, back to back on one box (directories solo_welch<N>_<random>); the first of each
for n in (4096, 1024):
    dirs = sorted(glob.glob(os.path.join(src, f"solo_welch{n}_*")), key=os.path.getmtime)
    for k, d in enumerate(dirs[:2]):
        f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
        if f:
            name = f"r06_solo_welch{'' if n == 4096 else n}_kernel_stats{'' if k == 0 else '_b'}.csv"
            shutil.copyfile(f[0], os.path.join(dst, name))
            print("  ", name)
# step timelines
for sub, name in (("dep_trace", "graph"), ("dep_trace_eager", "eager")):
    f = glob.glob(os.path.join(src, sub, "**", "*kernel_trace.csv"), recursive=True)
    if f:
        r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "step_timeline.py"), f[0]], capture_output=True, text=True)
        with open(os.path.join(dst, f"r06_deployment_timeline_{name}.txt"), "w") as out:
            out.write(f"# one steady-state step of the three-antenna 10-s deployment ({name}; tools/deployment_probe.py under rocprofv3 "
                      f"--kernel-trace, tools/step_timeline.py), commit {commit}\n" + r.stdout)
        print("  ", f"r06_deployment_timeline_{name}.txt")
f = glob.glob(os.path.join(src, "split8_trace", "**", "*kernel_trace.csv"), recursive=True)
if f:
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "split_timeline.py"), f[0]], capture_output=True, text=True)
    open(os.path.join(dst, "r06_split_emulated8_timeline.txt"), "w").write(r.stdout + r.stderr[-500:])
    print("   r06_split_emulated8_timeline.txt")
for d, tool, args in (("r06_pmc_welch", "pmc_summarize.py", [os.path.join(src, "pmc_welch4096"), commit, "4096"]),
                      ("r06_pmc_welch1024", "pmc_summarize.py", [os.path.join(src, "pmc_welch1024"), commit, "1024"]),
                      ("r06_pmc_xcorr", "pmc_family.py", [os.path.join(src, "pmc_sec"), "xcorr3", commit]),
                      ("r06_pmc_scan", "pmc_family.py", [os.path.join(src, "pmc_sec"), "cscan", commit])):
    os.makedirs(os.path.join(dst, d), exist_ok=True)
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", tool)] + args, capture_output=True, text=True)
    if r.returncode == 0 and r.stdout.strip().startswith("{"):
        open(os.path.join(dst, d, "summary.json"), "w").write(r.stdout)
        print("  ", d + "/summary.json")
    else:
        print("  FAILED", d, r.stderr[-300:])
