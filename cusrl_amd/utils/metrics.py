"""Count-weighted running means of recorded tensors (counterpart of cusrl/utils/metrics.py:12-96).

The reference updates a running mean per ``record`` call — ``mean()`` + ``mul_`` + ``mul`` + ``add_`` = four tiny
launches per metric per minibatch (≈ 700 launches per PPO update) — and reads every metric back with its own
``.item()``.  Here ``record`` only reduces the value to a 0-d device tensor (no launch at all for scalars such as the
losses) and queues it with its sample count; ``summary`` stacks the whole queue ONCE, copies it to the host ONCE and
forms the same count-weighted means there (in double).  Same numbers, ~2 launches per update instead of ~700.

Inside a captured hipGraph the queue cannot be used (graph-owned tensors are overwritten by the next replay), so a
:class:`MetricTap` collects the values of one captured step into a persistent accumulator instead (see
``cusrl_amd/template/graphs.py``).
"""

from __future__ import annotations

import itertools
from collections.abc import Mapping
from typing import Any

import torch

__all__ = ["Metric", "MetricTap", "Metrics", "StagedSummary"]


class Metric:
    """Resolved value of one metric (kept for API compatibility: ``metrics[name].mean`` / ``.count``)."""

    __slots__ = ("mean", "count")

    def __init__(self, mean: torch.Tensor | None = None, count: int = 0):
        self.mean = torch.tensor([]) if mean is None else mean
        self.count = count


class MetricTap:
    """Receives ``(name, 0-d tensor, count)`` while a step is being captured into a hipGraph."""

    def __init__(self, accumulator: torch.Tensor | None = None):
        self.names: list[str] = []
        self.values: list[torch.Tensor] = []
        self.counts: list[int] = []
        # the persistent running sums of the capture (slot i belongs to values[i]) and the slots whose value is added by the
        # launch that PRODUCES it (the flat Adam step adds its gradient norm itself): no accumulate launch for those
        self.accumulator = accumulator
        self.produced: set[int] = set()

    def add(self, name: str, value: torch.Tensor, count: int):
        self.names.append(name)
        self.values.append(value)
        self.counts.append(count)

    def slot_of(self, value: torch.Tensor) -> torch.Tensor | None:
        """The accumulator slot of a tapped value, handed to its producer: the producer adds the value itself (and the capture
        skips it).  None when the value is not tapped (or there is no accumulator)."""
        if self.accumulator is None:
            return None
        for index, held in enumerate(self.values):
            if held.data_ptr() == value.data_ptr() and index < self.accumulator.numel() and index not in self.produced:
                self.produced.add(index)
                return self.accumulator[index : index + 1]
        return None


class StagedSummary:
    """A metric summary whose values are on their way to the host (``Metrics.staged_summary``): ``resolve()`` waits for the copy
    and returns the ``{prefix + name: value}`` dict; until then the device may be given more work."""

    __slots__ = ("_finish", "_prefix", "_values")

    def __init__(self, finish, prefix: str):
        self._finish, self._prefix, self._values = finish, prefix, None

    def resolve(self) -> dict[str, float]:
        if self._values is None:
            self._values = {f"{self._prefix}{name}": metric.mean.item() for name, metric in self._finish().items()}
            self._finish = None
        return self._values


class Metrics:
    def __init__(self):
        self._queue: dict[str, list[tuple[torch.Tensor | float, int]]] = {}
        self._tap: MetricTap | None = None
        # device tensors of any length whose host values a callback turns into resolved contributions (the running sums the
        # captured hipGraphs keep): they travel to the host in the SAME copy as the queued scalars
        self._lazy: list[tuple[torch.Tensor, Any]] = []
        # objects with a `stage_metrics(metrics)` method that still hold un-staged device sums (captures that replayed since the
        # last read): asked right before the one host copy
        self._pending: list[Any] = []
        self._landings: dict[tuple, list] = {}  # pinned host buffers of the staged reads (_landing)

    # ------------------------------------------------------------------ recording
    @torch.no_grad()
    def record(self, metrics: Mapping[str, Any] | None = None, /, **kwargs: Any):
        """Record named values; each contributes ``value.mean()`` with weight ``value.numel()``."""
        for name, value in itertools.chain((metrics or {}).items(), kwargs.items()):
            if value is None:
                continue
            if not isinstance(value, torch.Tensor):
                try:
                    value = torch.as_tensor(value, dtype=torch.float32)
                except Exception as error:
                    raise ValueError(f"Failed to update metric '{name}'") from error
            numel = value.numel()
            if numel == 0:
                continue
            if numel > 1:
                value = value.float().mean() if value.dtype != torch.float32 else value.mean()
            else:
                value = value.detach().reshape(())
                if value.dtype != torch.float32:
                    value = value.float()
            if self._tap is not None:
                self._tap.add(name, value, numel)
            else:
                self._queue.setdefault(name, []).append((value, numel))

    def record_reduced(self, name: str, mean: torch.Tensor, count: int):
        """Record a value that is already the mean of ``count`` samples (0-d device tensor, e.g. a kernel output)."""
        if count <= 0:
            return
        mean = mean.detach().reshape(())
        if self._tap is not None:
            self._tap.add(name, mean, count)
        else:
            self._queue.setdefault(name, []).append((mean, count))

    def add_resolved(self, name: str, weighted_sum: float, count: int):
        """Merge an already reduced contribution (used by graph replays: Σ mean·count and Σ count)."""
        if count > 0:
            self._queue.setdefault(name, []).append((weighted_sum / count, count))

    def tap(self, tap: MetricTap | None):
        self._tap = tap

    def defer(self, values: torch.Tensor, callback) -> None:
        """``callback(list of floats)`` is called with the host values of the device tensor ``values`` when the metrics are next
        read; it records them with :meth:`add_resolved`."""
        self._lazy.append((values.reshape(-1), callback))

    def pending(self, source) -> None:
        """``source.stage_metrics(self)`` will be called before the next read (once)."""
        if not any(held is source for held in self._pending):
            self._pending.append(source)

    def clear(self):
        self._queue.clear()
        self._lazy.clear()

    # ------------------------------------------------------------------ reading
    def _resolve(self) -> dict[str, Metric]:
        return self._stage()()

    def _stage(self):
        """The device half of a read: everything recorded so far is concatenated ONCE per (device, dtype) and sent to the host in
        ONE copy each — into pinned memory, without waiting for it.  Returns the host half: a callable that waits for the copies
        (an event, not a stream synchronisation), runs the deferred callbacks and forms the count-weighted means.  Between the
        two the caller may enqueue whatever it likes (the trainer launches the next rollout, template/trainer.py): the store
        itself can be cleared and recorded into again, the staged read keeps its own references."""
        self._stage_pending()
        lazy, self._lazy = self._lazy, []
        queue = {name: list(entries) for name, entries in self._queue.items()}
        tensors: dict[tuple, list[torch.Tensor]] = {}
        for entries in queue.values():
            for value, _ in entries:
                if isinstance(value, torch.Tensor):
                    tensors.setdefault((value.device, value.dtype), []).append(value)
        for values, _ in lazy:
            tensors.setdefault((values.device, values.dtype), []).append(values)
        copies, events = [], []
        for (device, dtype), group in tensors.items():  # ONE concatenation + ONE host copy per (device, dtype), whatever was recorded
            flat = torch.cat([tensor.reshape(-1) for tensor in group]) if len(group) > 1 else group[0].reshape(-1)
            if device.type == "cuda":
                landing = self._landing(device, dtype, flat.numel())
                landing.copy_(flat, non_blocking=True)
                event = torch.cuda.Event()
                event.record(torch.cuda.current_stream(device))
                events.append(event)
                flat = landing
            copies.append((group, flat))

        def finish() -> dict[str, Metric]:
            for event in events:
                event.synchronize()
            host: dict[int, Any] = {}
            for group, flat in copies:
                values = flat.tolist()
                offset = 0
                for tensor in group:
                    n = tensor.numel()
                    host[id(tensor)] = values[offset] if tensor.dim() == 0 else values[offset : offset + n]
                    offset += n
            # (the callbacks record what they decode with `add_resolved`: into THIS read's entries — the store may have been
            # cleared and recorded into again since the read was staged)
            live, self._queue = self._queue, queue
            try:
                for values, callback in lazy:
                    callback(host[id(values)])
            finally:
                self._queue = live
            resolved = {}
            for name, entries in queue.items():
                total = sum(count for _, count in entries)
                mean = sum((host[id(v)] if isinstance(v, torch.Tensor) else v) * (count / total) for v, count in entries)
                resolved[name] = Metric(torch.tensor(mean, dtype=torch.float32), total)
            return resolved

        return finish

    def _landing(self, device, dtype, numel: int) -> torch.Tensor:
        """Pinned host memory for one staged copy: two buffers per (device, dtype) used in turn, so that a read staged while the
        previous one has not been finished yet lands somewhere else."""
        ring = self._landings.setdefault((device, dtype), [None, None, 0])
        slot = ring[2] = ring[2] ^ 1
        if ring[slot] is None or ring[slot].numel() < numel:
            ring[slot] = torch.empty(max(numel, 256), dtype=dtype).pin_memory()
        return ring[slot][:numel]

    def _stage_pending(self):
        """Snapshot (ONE concatenation) and reset (ONE multi-tensor fill) the device-side running sums of every capture that
        replayed since the last read, whatever their number — the per-capture ``tolist`` of rounds 2-5 was one host
        synchronisation per captured graph and update (nine for the ``ppo`` preset)."""
        sources, self._pending = self._pending, []
        staged = []
        for source in sources:
            staged.extend(source.stage_metrics(self) or ())
        staged = [(tensor, reset, callback) for tensor, reset, callback in staged if tensor.numel()]
        by_type: dict[tuple, list] = {}
        for entry in staged:
            by_type.setdefault((entry[0].device, entry[0].dtype), []).append(entry)
        for group in by_type.values():
            snapshot = torch.cat([tensor.reshape(-1) for tensor, _, _ in group])
            resets = [reset for _, reset, _ in group if reset is not None]  # (None: a piece that is only read)
            if resets:
                torch._foreach_zero_(resets)
            sizes = [tensor.numel() for tensor, _, _ in group]
            callbacks = [callback for _, _, callback in group]

            def scatter(values, sizes=sizes, callbacks=callbacks):
                offset = 0
                for size, callback in zip(sizes, callbacks):
                    callback(values[offset : offset + size])
                    offset += size

            self.defer(snapshot, scatter)

    def summary(self, prefix: str = "") -> dict[str, float]:
        return self.staged_summary(prefix).resolve()

    def staged_summary(self, prefix: str = "") -> "StagedSummary":
        """:meth:`summary` in two halves: the device copies are issued now, the host values are formed by ``.resolve()``."""
        if prefix and not prefix.endswith("/"):
            prefix += "/"
        return StagedSummary(self._stage(), prefix)

    def __getitem__(self, name: str) -> Metric:
        return self._resolve()[name]

    def __iter__(self):
        return iter(self._queue)

    def __len__(self):
        return len(self._queue)

    def keys(self):
        return self._queue.keys()

    def items(self):
        return self._resolve().items()

    def values(self):
        return self._resolve().values()

    def get(self, name: str, default=None):
        return self._resolve().get(name, default)
