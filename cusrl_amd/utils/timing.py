"""Named section timers (counterpart of cusrl/utils/timing.py:97-179): wall clock on CPU, HIP events on a GPU."""

from __future__ import annotations

import time
from collections import defaultdict
from contextlib import contextmanager

import torch

from cusrl_amd.utils.config import device as resolve_device

__all__ = ["Timer"]


class Timer:
    def __init__(self, device: torch.device | str | None = None):
        self.device = resolve_device(device)
        self._gpu = self.device.type == "cuda"
        self._open: dict[str, object] = {}
        self._total: dict[str, float] = defaultdict(float)
        self._pending: dict[str, list] = defaultdict(list)

    def _now(self):
        if not self._gpu:
            return time.perf_counter()
        event = torch.cuda.Event(enable_timing=True)
        event.record(torch.cuda.current_stream(self.device))
        return event

    def start(self, name):
        if name in self._open:
            raise RuntimeError(f"Timer '{name}' has already been started")
        self._open[name] = self._now()

    def stop(self, name):
        if name not in self._open:
            raise RuntimeError(f"Timer '{name}' has not been started")
        begin = self._open.pop(name)
        if self._gpu:
            self._pending[name].append((begin, self._now()))  # resolved lazily: no sync inside the loop
        else:
            self._total[name] += time.perf_counter() - begin

    def __getitem__(self, name) -> float:
        for begin, end in self._pending.pop(name, []):
            end.synchronize()
            self._total[name] += begin.elapsed_time(end) / 1000.0
        return self._total[name]

    def clear(self):
        self._open.clear()
        self._total.clear()
        self._pending.clear()

    @contextmanager
    def record(self, name):
        self.start(name)
        yield
        self.stop(name)
