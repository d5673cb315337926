#!/usr/bin/env python3
"""The three-antenna 10-s deployment step of bench.py (`deployment`) alone -- the thing to put after `rocprofv3 ... --`
for a kernel trace of the reference's own operating point.
    python tools/deployment_probe.py"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "gps-jamming_amd"))


def main():
    import numpy as np
    import torch
    import gpsjam
    import bench
    from gpsjam.synth import StreamSpec
    dev = gpsjam.Device(0)
    ws = torch.cuda.Stream()
    torch.cuda.set_stream(ws)
    dev.set_stream(ws.cuda_stream)
    graph = "--eager" not in sys.argv
    kw = {"scan_first": True} if "--scan-first" in sys.argv else {}
    r = bench.deployment(np, torch, gpsjam, dev, StreamSpec, graph=graph, **kw)["line"]
    print("graph" if graph else "eager", {k: r[k] for k in ("resident_step_ms", "resident_step_latency_ms", "file_to_results_ms")})


if __name__ == "__main__":
    main()
