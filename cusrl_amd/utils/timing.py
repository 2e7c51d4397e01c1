"""Named section timers (counterpart of cusrl/utils/timing.py:97-179): wall clock on CPU, HIP events on a GPU."""

from __future__ import annotations

import time
from collections import defaultdict
from contextlib import contextmanager

import torch

from cusrl_amd.utils.config import device as resolve_device

__all__ = ["Timer"]


class Timer:
    """Accumulates time per named section.  On a GPU the sections are bracketed by HIP timing events resolved lazily
    (no synchronisation inside the loop).  Timing events are not free on this stack — each one is a barrier packet
    that costs the stream a bubble — so back-to-back sections SHARE their boundary event (the end of one section is
    the start of the next when it begins within 50 us of host time): the trainer's 4 sections per env step cost
    4 events instead of 8.  ``record(name, every=k)`` additionally SAMPLES a repetitive section: only every k-th
    occurrence is bracketed and the accumulated time is scaled by occurrences / bracketed occurrences — the env-step
    sections of the rollout loop cost ~0.6 ms per iteration (5 %) when every one of them is timed."""

    SHARE_WINDOW = 50e-6

    def __init__(self, device: torch.device | str | None = None):
        self.device = resolve_device(device)
        self._gpu = self.device.type == "cuda"
        self._open: dict[str, object] = {}
        self._total: dict[str, float] = defaultdict(float)
        self._pending: dict[str, list] = defaultdict(list)
        self._boundary = None
        self._boundary_time = 0.0
        self._seen: dict[tuple, int] = defaultdict(int)   # (name, every) -> occurrences
        self._timed: dict[tuple, int] = defaultdict(int)  # (name, every) -> occurrences that were bracketed

    def _event(self):
        event = torch.cuda.Event(enable_timing=True)
        event.record(torch.cuda.current_stream(self.device))
        return event

    def start(self, name):
        if name in self._open:
            raise RuntimeError(f"Timer '{name}' has already been started")
        if not self._gpu:
            self._open[name] = time.perf_counter()
        elif self._boundary is not None and time.perf_counter() - self._boundary_time < self.SHARE_WINDOW:
            self._open[name] = self._boundary
        else:
            self._open[name] = self._event()

    def stop(self, name):
        if name not in self._open:
            raise RuntimeError(f"Timer '{name}' has not been started")
        begin = self._open.pop(name)
        if self._gpu:
            end = self._event()
            self._pending[name].append((begin, end))  # resolved lazily: no sync inside the loop
            self._boundary, self._boundary_time = end, time.perf_counter()
        else:
            self._total[name] += time.perf_counter() - begin

    def _resolve(self, name):
        for begin, end in self._pending.pop(name, []):
            end.synchronize()
            self._total[name] += begin.elapsed_time(end) / 1000.0

    def __getitem__(self, name) -> float:
        self._resolve(name)
        total = self._total[name]
        for key, seen in self._seen.items():  # sampled sections of this name, scaled to all their occurrences
            if key[0] == name and self._timed[key]:
                self._resolve(key)
                total += self._total[key] * (seen / self._timed[key])
        return total

    def detach(self) -> "Timer":
        """What this timer has accumulated so far, as a timer of its own (to be read later — e.g. after sections of the NEXT
        iteration have already been recorded into this one, template/trainer.py); this timer starts again from zero."""
        if self._open:
            raise RuntimeError(f"Timer sections still open: {sorted(map(str, self._open))}")
        frozen = Timer.__new__(Timer)
        frozen.device, frozen._gpu = self.device, self._gpu
        frozen._open, frozen._boundary, frozen._boundary_time = {}, None, 0.0
        frozen._total, frozen._pending, frozen._seen, frozen._timed = self._total, self._pending, self._seen, self._timed
        self._total, self._pending = defaultdict(float), defaultdict(list)
        self._seen, self._timed = defaultdict(int), defaultdict(int)
        self._boundary = None
        return frozen

    def clear(self):
        self._pending.clear()
        self._boundary = None
        self._open.clear()
        self._total.clear()
        self._seen.clear()
        self._timed.clear()

    @contextmanager
    def record(self, name, every: int = 1):
        if every > 1:
            key = (name, every)
            self._seen[key] += 1
            if (self._seen[key] - 1) % every:
                yield
                return
            self._timed[key] += 1
            name = key
        self.start(name)
        yield
        self.stop(name)
