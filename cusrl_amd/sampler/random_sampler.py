"""Uniform random sampling from the valid region of a (possibly still filling, possibly wrapped) buffer
(cusrl/sampler/random_sampler.py:18-138) on the one-launch gather kernel.

Index semantics are the reference's: ``RandomSampler`` draws ``batch_size`` flat slots with ``torch.randint`` over the
valid prefix (``cursor`` rows while the ring is filling, all of it once full); ``TemporalRandomSampler`` draws one env id
and one logical start step per sequence (env ids first, starts second — the order of the two ``randint`` calls is part of
the random stream) and reads ``sequence_len`` consecutive steps, logical time 0 being ``cursor`` once the ring is full.
What differs is the data movement: the window ``data[time_indices, env_indices]`` is one flat slot list
(``cusrl_window_indices``: ``((cursor + start + t) % capacity) * parallelism + env``) fed to the same multi-leaf gather
launch the minibatch samplers use, and batches are lazy (only the fields a consumer reads are moved).
"""

from __future__ import annotations

from typing import Any

import torch

from cusrl_amd import ops
from cusrl_amd.template.buffer import Buffer, Sampler

__all__ = ["AutoRandomSampler", "RandomSampler", "TemporalRandomSampler"]


class _RandomBase(Sampler):
    def __init__(self, num_batches: int, batch_size: int, *, index_device: str | torch.device | None = None, lazy: bool = True):
        self.num_batches = num_batches
        self.batch_size = batch_size
        # where torch.randint draws (default: the buffer's device, like the reference); "cpu" reproduces a CPU
        # reference's stream on a GPU buffer (indices are then uploaded)
        self.index_device = None if index_device is None else torch.device(index_device)
        self.lazy = lazy
        self.hot_fields: set[str] = set()

    def _randint(self, high: int, buffer: Buffer) -> torch.Tensor:
        device = self.index_device or buffer.device
        drawn = torch.randint(high, (self.batch_size,), device=device)
        return drawn if device == buffer.device else drawn.to(buffer.device, non_blocking=True)

    def _prepare(self, buffer: Buffer, rows_per_batch: int):
        """The per-slot record only pays when a pass samples about as many rows as the buffer holds: refreshing it costs
        a pass over every leaf that changed since the last one — after a single ``push`` that is the whole narrow part of
        the buffer, far more than a replay-style consumer (push a step, sample a small batch) reads.  Below that the
        gather simply reads the leaves (anything the record still mirrors is used as it is)."""
        if self.num_batches * rows_per_batch >= buffer.capacity * buffer.get_parallelism():
            buffer.prepare_sampling(self.hot_fields if self.lazy else None)

    def _batch(self, buffer: Buffer, slots: torch.Tensor, lead_shape: tuple[int, ...] | None):
        if self.lazy:
            return buffer.gather_lazy(slots, False, self.hot_fields, lead_shape=lead_shape)
        return buffer.gather(slots, lead_shape=lead_shape)


class RandomSampler(_RandomBase):
    """Independent transitions, uniformly from the valid region (``:18-62``)."""

    def __call__(self, buffer: Buffer):
        num_samples = (buffer.capacity if buffer.full else buffer.cursor) * buffer.get_parallelism()
        self._prepare(buffer, self.batch_size)
        previous = None
        for batch_index in range(self.num_batches):
            metadata = {"batch_index": batch_index, "total_batches": self.num_batches, "temporal": False}
            # the valid prefix is the first `cursor` rows of every [capacity, parallelism, ...] leaf, so a flat index
            # into the prefix is a flat slot index of the whole leaf
            slots = self._randint(num_samples, buffer)
            if previous is not None:
                previous.expire()
            batch = self._batch(buffer, slots, None)
            previous = batch if self.lazy else None
            yield metadata, batch
        if previous is not None:
            previous.expire()


class TemporalRandomSampler(_RandomBase):
    """Random temporal windows, each with its own env id and start step (``:65-113``)."""

    def __init__(self, num_batches: int, batch_size: int, sequence_len: int | None = None, *,
                 index_device: str | torch.device | None = None, lazy: bool = True):
        if sequence_len is not None and sequence_len <= 0:
            raise ValueError("'sequence_len' must be positive or None")
        super().__init__(num_batches, batch_size, index_device=index_device, lazy=lazy)
        self.sequence_len = sequence_len

    def __call__(self, buffer: Buffer):
        full, cursor = buffer.full, buffer.cursor
        valid_sequence_len = buffer.capacity if full else cursor
        sequence_len = valid_sequence_len if self.sequence_len is None else min(self.sequence_len, valid_sequence_len)
        if sequence_len == 0:
            raise RuntimeError("TemporalRandomSampler can sample only from a non-empty buffer")
        num_starts = valid_sequence_len - sequence_len + 1  # starts live in logical time (oldest valid step = 0)
        self._prepare(buffer, self.batch_size * sequence_len)
        previous = None
        for batch_index in range(self.num_batches):
            metadata = {"batch_index": batch_index, "total_batches": self.num_batches, "temporal": True}
            env_indices = self._randint(buffer.get_parallelism(), buffer)
            start_indices = self._randint(num_starts, buffer)
            slots = ops.window_indices(start_indices, env_indices, sequence_len, buffer.capacity, buffer.get_parallelism(),
                                       cursor if full else None)
            if previous is not None:
                previous.expire()
            batch = self._batch(buffer, slots, (sequence_len, self.batch_size))
            previous = batch if self.lazy else None
            yield metadata, batch
        if previous is not None:
            previous.expire()


class AutoRandomSampler(Sampler):
    """Temporal windows iff some top-level field name ends with ``memory`` (``:116-138``)."""

    def __init__(self, num_batches: int, batch_size: int, sequence_len: int | None = None, *,
                 index_device: str | torch.device | None = None, lazy: bool = True):
        self.num_batches = num_batches
        self.batch_size = batch_size
        self.sequence_len = sequence_len
        self.index_device = index_device
        self.lazy = lazy
        self.hot_fields: set[str] = set()

    def __call__(self, buffer: Buffer):
        temporal = any(key.split(".")[0].endswith("memory") for key in buffer)
        extra: dict[str, Any] = {"index_device": self.index_device, "lazy": self.lazy}
        if temporal:
            sampler: _RandomBase = TemporalRandomSampler(self.num_batches, self.batch_size, self.sequence_len, **extra)
        else:
            sampler = RandomSampler(self.num_batches, self.batch_size, **extra)
        sampler.hot_fields = self.hot_fields
        return sampler(buffer)
