"""HIP streams that really run side by side.

The HIP runtime maps streams onto a small pool of hardware queues (``GPU_MAX_HW_QUEUES``, four by default; a new stream
gets the least-used queue) and two streams on ONE queue execute one after the other.  The pipelines here keep K2 on one
stream and a chain of small kernels on a second (and, on the combining rank of a split run, a third) precisely so that
they overlap -- and the kernel traces of round 4 showed both failure modes: a third stream on the main stream's queue
(K2 queued behind the combine), and, with eight queues, the deployment step's second stream on its first one's (the step
took 1.13 instead of 0.47 ms).  Which queue a stream lands on depends on every stream the process has made before, so it
cannot be arranged; it can be TESTED: keep one stream busy with a spinning wave (gj_debug_busy_dev) and see whether an
event recorded on the other completes meanwhile.
"""
from __future__ import annotations

import logging
import time
from typing import Sequence, Tuple

import torch

_log = logging.getLogger("gpsjam.streams")


def runs_beside(dev, busy_stream, other_stream, busy_ms: float = 2.0) -> bool:
    """True when work on ``other_stream`` completes while ``busy_stream`` (the stream ``dev`` is bound to) is occupied:
    the two do not share a hardware queue."""
    torch.cuda.synchronize()
    done_busy, done_other = torch.cuda.Event(), torch.cuda.Event()
    dev.debug_busy_dev(busy_ms)
    done_busy.record(busy_stream)
    done_other.record(other_stream)
    t0 = time.perf_counter()
    while not done_other.query() and time.perf_counter() - t0 < busy_ms * 4e-3:
        pass
    beside = bool(done_other.query()) and not done_busy.query()
    torch.cuda.synchronize()
    return beside


def stream_beside(against: Sequence[Tuple[object, "torch.cuda.Stream"]], device=None, priority: int = 0, tries: int = 8):
    """A new stream that runs side by side with every stream of ``against`` ([(Device bound to it, stream), ...]).
    Candidates that share a queue with one of them are kept alive until the search ends (so the next candidate is dealt
    another queue) and dropped afterwards.  If none of ``tries`` candidates qualifies the last one is returned and a
    warning logged: the pipeline is still correct, its streams just do not overlap."""
    rejected, cand = [], None
    for _ in range(max(1, tries)):
        cand = torch.cuda.Stream(device=device, priority=priority)
        if all(runs_beside(dev, s, cand) for dev, s in against):
            break
        rejected.append(cand)
    else:
        _log.warning("no stream found that runs beside the pipeline's other streams in %d tries (GPU_MAX_HW_QUEUES=%s?): "
                     "the side chain will run behind K2 instead of beside it", tries, __import__("os").environ.get("GPU_MAX_HW_QUEUES", "4"))
    del rejected
    return cand
