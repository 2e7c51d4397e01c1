"""Side streams that really run beside the stream they are meant to overlap with.

HIP multiplexes a process's streams onto a handful of hardware queues (four by default), and work of two streams that share a
queue runs in stream-issue order — serially.  Which queue a new stream lands on depends on how many streams the process has
created before it: in a one-process run the sampler's draw-ahead stream happened to get a queue of its own, in a torchrun rank
(process-group streams, RCCL's) it shared the MAIN stream's queue, and the six ``randperm`` draws of an update — issued to
run under ``pre_update``'s head region — sat between that region and its tail instead: +0.3 ms per iteration
(profiles/r06/one_rank/pre_update_one_rank_before.txt).  So a side stream is not taken on faith: candidates are created until
one demonstrably makes progress while the streams it has to run beside are busy.  (The reference has no counterpart: it issues
everything on one stream, cusrl/template/actor_critic.py:296-320.)
"""

from __future__ import annotations

import os
import time
from collections.abc import Sequence

import torch

__all__ = ["runs_beside", "side_stream"]

_SPIN_CYCLES = 2_000_000  # torch.cuda._sleep: ~0.8 ms on this part — long against an event's submission AND against a host thread that is descheduled for a moment, short against a start-up
_CANDIDATES = 12


def runs_beside(candidate: "torch.cuda.Stream", busy: Sequence["torch.cuda.Stream"]) -> bool:
    """Does work issued to ``candidate`` make progress while every stream of ``busy`` is occupied?  Each busy stream is handed a
    spin kernel; an event recorded on the candidate right behind them must complete before the spins do."""
    device = candidate.device
    with torch.cuda.device(device):
        torch.cuda.synchronize(device)
        spun = []
        for stream in busy:
            with torch.cuda.stream(stream):
                torch.cuda._sleep(_SPIN_CYCLES)
                done = torch.cuda.Event()
                done.record(stream)
                spun.append(done)
        with torch.cuda.stream(candidate):
            marker = torch.cuda.Event()
            marker.record(candidate)
        deadline = time.perf_counter() + 0.05
        beside = False
        while time.perf_counter() < deadline:
            if marker.query():
                beside = not all(done.query() for done in spun)  # (reached while a spin was still running)
                break
            if all(done.query() for done in spun):
                break
        torch.cuda.synchronize(device)
    return beside


def side_stream(device: torch.device, beside: Sequence["torch.cuda.Stream"] | None = None) -> "torch.cuda.Stream":
    """A new stream on ``device`` whose work overlaps with the work of ``beside`` (default: the current stream).  Falls back to
    the last candidate when none passes (a part with a single queue: correct, just serial).  Never call while capturing."""
    device = torch.device(device)
    busy = list(beside) if beside is not None else [torch.cuda.current_stream(device)]
    if torch.cuda.is_current_stream_capturing() or os.environ.get("CUSRL_SIDE_STREAM_PROBE", "1") == "0":  # (A/B switch)
        return torch.cuda.Stream(device=device)
    # (Not a high-priority stream, although the runtime keeps a pool of hardware queues per priority and such a stream could share a
    # queue with nothing else here: with the six randperm draws of an update on one, the single-process iteration went from 4.8 to
    # 10.3 ms and a one-rank iteration from 5.0 to 7.2 ms on this stack — profiles/r06/experiments/side_stream_priority_ab.txt.)
    high = os.environ.get("CUSRL_SIDE_STREAM_PRIORITY", "0") == "1"  # (A/B switch, off: see above)
    candidate = torch.cuda.Stream(device=device, priority=-1) if high else torch.cuda.Stream(device=device)
    rejected = []  # (kept alive until the choice is made: a released stream would be handed out again)
    for _ in range(_CANDIDATES):
        # twice: a queue that is merely still starting up must not look like a shared one
        if runs_beside(candidate, busy) or runs_beside(candidate, busy):
            return candidate
        rejected.append(candidate)
        candidate = torch.cuda.Stream(device=device)
    return candidate
