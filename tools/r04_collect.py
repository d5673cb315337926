#!/usr/bin/env python3
"""Copy the judged evidence of a `tools/r04_profile.sh <tag>` run from gpurun_out/<tag>/ into profiles/ (tracked):
    python tools/r04_collect.py r04c"""
import glob
import os
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04c"
src = os.path.join(REPO, "gpurun_out", tag)
dst = os.path.join(REPO, "profiles")
commit = subprocess.run(["git", "-C", REPO, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()


def cp(a, b):
    a = os.path.join(src, a)
    if os.path.exists(a):
        shutil.copyfile(a, os.path.join(dst, b))
        print("  ", b)
    else:
        print("  MISSING", a)


for a, b in (("bench.json", "r04_bench.json"), ("force_exchange.json", "r04_force_exchange_n1.json"),
             ("weak_emulated8.json", "r04_weak_emulated8.json"), ("split_n1.json", "r04_split_n1.json"),
             ("split_emulated8.json", "r04_split_emulated8_rank0.json"), ("split_emulated4.json", "r04_split_emulated4_rank0.json"),
             ("split_emulated8_rank1.json", "r04_split_emulated8_rank1.json"), ("split_emulated8_rank2.json", "r04_split_emulated8_rank2.json"),
             ("split_emulated8_rank7.json", "r04_split_emulated8_rank7.json"), ("split_n2_share.json", "r04_split_n2_share_gpu.json"),
             ("weak_n2_share.json", "r04_weak_n2_share_gpu.json")):
    cp(a, b)
for sub, name in (("bench_trace", "r04_bench_kernel_stats.csv"), ("split8_trace", "r04_split_emulated8_kernel_stats.csv"),
                  ("acq_trace", "r04_acq_kernel_stats.csv"), ("pmc_welch1024/trace", "r04_solo_welch1024_kernel_stats.csv"),
                  ("pmc_welch4096/trace", "r04_solo_welch_kernel_stats.csv")):
    f = glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)
    if f:
        shutil.copyfile(f[0], os.path.join(dst, name))
        print("  ", name)
    else:
        print("  MISSING kernel stats under", sub)
with open(os.path.join(dst, "r04_deployment_probe.txt"), "w") as out:
    for n in ("deployment_graph.txt", "deployment_eager.txt"):
        p = os.path.join(src, n)
        if os.path.exists(p):
            out.write("".join(ln for ln in open(p) if "amdgpu.ids" not in ln))
for d, tool, args in (("r04_pmc_welch", "pmc_summarize.py", [os.path.join(src, "pmc_welch4096"), commit, "4096"]),
                      ("r04_pmc_welch1024", "pmc_summarize.py", [os.path.join(src, "pmc_welch1024"), commit, "1024"]),
                      ("r04_pmc_acq", "pmc_family.py", [os.path.join(src, "pmc_sec"), "acq", commit]),
                      ("r04_pmc_scan", "pmc_family.py", [os.path.join(src, "pmc_sec"), "fscan", commit])):
    os.makedirs(os.path.join(dst, d), exist_ok=True)
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", tool)] + args, capture_output=True, text=True)
    if r.returncode == 0 and r.stdout.strip().startswith("{"):
        open(os.path.join(dst, d, "summary.json"), "w").write(r.stdout)
        print("  ", d + "/summary.json")
    else:
        print("  FAILED", d, r.stderr[-300:])
