"""HIP streams that really run side by side.

The HIP runtime maps streams onto a small pool of hardware queues (``GPU_MAX_HW_QUEUES``, four by default; a new stream
gets the least-used queue) and two streams on ONE queue execute one after the other.  The pipelines here keep K2 on one
stream and a chain of small kernels on a second (and, on the combining rank of a split run, a third) precisely so that
they overlap -- and the kernel traces of round 4 showed both failure modes: a third stream on the main stream's queue
(K2 queued behind the combine), and, with eight queues, the deployment step's second stream on its first one's (the step
took 1.13 instead of 0.47 ms).  Which queue a stream lands on depends on every stream the process has made before, so it
cannot be arranged; it can be TESTED: keep one stream busy with a spinning wave (gj_probe_busy_dev) and see whether an
event recorded on the other completes meanwhile.
"""
from __future__ import annotations

import logging
import os
from typing import Sequence, Tuple

import torch

_log = logging.getLogger("gpsjam.streams")


def runs_beside(dev, busy_stream, other_stream, busy_ms: float = 2.0) -> bool:
    """True when work on ``other_stream`` completes while ``busy_stream`` (the stream ``dev`` is bound to) is occupied:
    the two do not share a hardware queue.

    Only the two streams involved are synchronised (a pipeline in mid-step on other streams of the device is not
    stalled), and the verdict is read from the GPU's own time stamps of two events -- ``other`` finished well before
    ``busy`` -- rather than from host polling, so a descheduled host thread cannot turn it."""
    busy_stream.synchronize()
    other_stream.synchronize()
    done_busy, done_other = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dev.probe_busy_dev(busy_ms)                   # gj_probe_busy_dev: one spinning wave on the context's stream
    done_busy.record(busy_stream)
    done_other.record(other_stream)
    done_other.synchronize()
    done_busy.synchronize()
    return done_other.elapsed_time(done_busy) > 0.5 * busy_ms


def stream_beside_checked(against: Sequence[Tuple[object, "torch.cuda.Stream"]], device=None, priority: int = 0, tries: int = 8):
    """(stream, overlaps): a new stream and whether it runs side by side with every stream of ``against``
    ([(Device bound to it, stream), ...]).  Candidates that share a queue with one of them are kept alive until the
    search ends (so the next candidate is dealt another queue) and dropped afterwards.  If none of ``tries`` candidates
    qualifies the last one is returned with ``overlaps`` False and a warning logged: the pipeline is still correct (its
    cross-stream ordering is by events), its streams just do not overlap.  Inside a stream capture nothing can be
    probed (a synchronisation would invalidate the capture): the candidate is returned untested, with a warning."""
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        _log.warning("stream_beside called inside a stream capture: the new stream is not tested for overlap")
        return torch.cuda.Stream(device=device, priority=priority), False
    rejected, cand, ok = [], None, False
    for _ in range(max(1, tries)):
        cand = torch.cuda.Stream(device=device, priority=priority)
        if all(runs_beside(dev, s, cand) for dev, s in against):
            ok = True
            break
        rejected.append(cand)
    else:
        _log.warning("no stream found that runs beside the pipeline's other streams in %d tries (GPU_MAX_HW_QUEUES=%s?): "
                     "the side chain will run behind K2 instead of beside it", tries, os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    del rejected
    return cand, ok


def stream_beside(against, device=None, priority: int = 0, tries: int = 8):
    """The stream of :func:`stream_beside_checked` alone."""
    return stream_beside_checked(against, device=device, priority=priority, tries=tries)[0]
