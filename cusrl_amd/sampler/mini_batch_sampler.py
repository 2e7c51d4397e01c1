"""Shuffled minibatch iteration over a full buffer (cusrl/sampler/mini_batch_sampler.py:12-140).

Index semantics are the reference's, bit for bit: one ``torch.randperm(S)`` from the GLOBAL generator of the
permutation device per call, re-drawn in place (``out=``) for every later epoch when ``shuffle``;
``mini_batch_size = S // num_mini_batches`` (floor — tail indices are dropped); minibatch ``j`` is the slice
``[j*B, (j+1)*B)``; flat index ``i`` addresses slot ``(t, n) = (i // N, i % N)``.  The permutation device defaults
to the buffer's device, exactly like the reference (``device=buffer.device``), so a GPU run consumes torch's
Philox stream and a CPU-generator run (``permutation_device="cpu"``) reproduces the CPU reference's mt19937
stream on a GPU buffer (indices are then uploaded, 8 B per sample).  The gather of all leaves is one HIP launch.
"""

from __future__ import annotations

import os
from collections.abc import Sequence

import torch

from cusrl_amd.template.buffer import Buffer, Sampler

__all__ = ["AutoMiniBatchSampler", "DrawnEpochs", "MiniBatchSampler", "TemporalMiniBatchSampler"]


_PREFETCH_STREAMS: dict[torch.device, torch.cuda.Stream] = {}


def _prefetch_stream(device: torch.device) -> torch.cuda.Stream:
    stream = _PREFETCH_STREAMS.get(device)
    if stream is None:
        # (a stream that demonstrably runs beside the stream the loop issues on: utils/streams.py)
        from cusrl_amd.utils.streams import side_stream

        stream = _PREFETCH_STREAMS[device] = side_stream(device)
    return stream


class DrawnEpochs:
    """The permutations of one pass over the buffer, one row of ``permutations [E, S]`` per epoch, drawn in epoch order on the
    draw-ahead stream when asked for: ``draw(e)`` issues epoch e's ``randperm`` (idempotent; epochs in front of it first),
    ``wait(e)`` makes the current stream wait for it.  A consumer that launches epoch e's steps and THEN draws epoch e + 1 has
    the draw's dozen launches run under those steps instead of in front of the first one; the generator still sees one
    ``randperm`` per epoch in epoch order (cusrl/sampler/mini_batch_sampler.py:56,67-68)."""

    def __init__(self, permutations: torch.Tensor, plan: list, stream: "torch.cuda.Stream"):
        self.permutations, self.plan, self.stream = permutations, plan, stream
        self.events: list[torch.cuda.Event] = []

    def __len__(self) -> int:
        return len(self.plan)

    def draw(self, epoch: int) -> None:
        while len(self.events) <= epoch < len(self.plan):
            row = self.permutations[len(self.events)]
            with torch.cuda.stream(self.stream):
                torch.randperm(row.numel(), device=row.device, out=row)
                event = torch.cuda.Event()
                event.record(self.stream)
            self.events.append(event)

    def wait(self, epoch: int) -> None:
        self.draw(epoch)
        torch.cuda.current_stream(self.permutations.device).wait_event(self.events[epoch])


class MiniBatchSampler(Sampler):
    temporal = False

    def __init__(
        self,
        num_epochs: int = 1,
        num_mini_batches: int | Sequence[int] = 1,
        shuffle: bool = True,
        *,
        permutation_device: str | torch.device | None = None,
        lazy: bool = True,
        prefetch: bool = True,
    ):
        if num_epochs <= 0:
            raise ValueError("'num_epochs' must be positive")
        self.num_epochs = num_epochs
        if isinstance(num_mini_batches, int):
            if num_mini_batches <= 0:
                raise ValueError("'num_mini_batches' must be positive")
            self.num_mini_batches: int | tuple[int, ...] = num_mini_batches
        else:
            self.num_mini_batches = tuple(num_mini_batches)
            if len(self.num_mini_batches) != num_epochs:
                raise ValueError(
                    "'num_mini_batches' must be an integer or a sequence of integers with length "
                    f"equal to 'num_epochs' ({num_epochs}); got {len(self.num_mini_batches)} values"
                )
            if any(v <= 0 for v in self.num_mini_batches):
                raise ValueError("'num_mini_batches' values must be positive")
        self.shuffle = shuffle
        self.permutation_device = None if permutation_device is None else torch.device(permutation_device)
        # lazy (extension, default): batches are LazyBatch dicts — every field of the buffer is available, only the ones
        # somebody reads are gathered (template/buffer.py); lazy=False gathers every leaf up front like the reference
        self.lazy = lazy
        self.hot_fields: set[str] = set()
        # prefetch (extension, default): a device permutation for the next epoch is drawn on a second stream while the
        # current epoch's minibatches run (same generator stream of values; see iter_indices)
        self.prefetch = prefetch and os.environ.get("CUSRL_PREFETCH_PERMUTATIONS", "1") != "0"
        # Device permutations are drawn into PERSISTENT index buffers (two, alternating when drawing ahead): every
        # minibatch's index slice then lives at an address that repeats from update to update, and a captured minibatch
        # step can read it in place instead of through a copy into a static buffer (`persistent_indices`, template/graphs.py).
        # Same generator calls, same values: `randperm(n, out=buffer)` is what `randperm(n, device=...)` runs on a fresh tensor.
        self._index_buffers: dict[tuple, list[torch.Tensor]] = {}
        self.persistent_indices = False

    def iter_indices(self, buffer: Buffer):
        """Yield ``(metadata, device index slice)`` per minibatch — the permutation logic of ``__call__`` without
        the gather, so a captured hipGraph can own the gather (template/graphs.py)."""
        if not (buffer.full and buffer.cursor == 0):
            raise RuntimeError("MiniBatchSampler requires a full buffer with cursor reset to 0")
        num_samples = self._get_num_samples(buffer)
        # the per-slot record, once per pass (a flag check when it is current): the fields earlier passes read, if known
        buffer.prepare_sampling(self.hot_fields if self.lazy else None)
        perm_device = self.permutation_device or buffer.device
        staged = perm_device != buffer.device
        persistent = not staged and perm_device.type == "cuda" and not torch.cuda.is_current_stream_capturing()
        self.persistent_indices = persistent
        if persistent:
            pair = self._index_buffers.get((num_samples, perm_device))
            if pair is None:
                pair = self._index_buffers[(num_samples, perm_device)] = [
                    torch.empty(num_samples, dtype=torch.int64, device=perm_device) for _ in range(2)]
            elif self.prefetch:
                # the buffers outlive a pass: should the previous pass have been abandoned mid-epoch (an exception, a consumer
                # stopping early) a draw-ahead may still be writing one of them on the side stream — order this draw behind it
                torch.cuda.current_stream(perm_device).wait_stream(_prefetch_stream(perm_device))
            epoch_indices = torch.randperm(num_samples, device=perm_device, out=pair[0])
        else:
            epoch_indices = torch.randperm(num_samples, device=perm_device)
        device_indices = epoch_indices.to(buffer.device, non_blocking=True) if staged else epoch_indices
        # A device permutation is a dozen launches (keys, radix / merge sort, de-duplication: ~60 us of device time) in
        # front of an epoch whose minibatch steps are device-bound.  The permutation of epoch e + 1 depends on nothing
        # epoch e computes, so it is drawn at the START of epoch e on a second stream, into the other of two index
        # buffers; the main stream picks it up with one event wait.  The generator is still asked for one randperm per
        # epoch in epoch order; the only thing that moves is that draw relative to random numbers the minibatch steps
        # themselves consume, which is why ActorCritic switches `prefetch` off when a hook or a dropout layer draws
        # inside the steps (then the interleaving is the reference's).
        ahead = (self.prefetch and self.shuffle and not staged and epoch_indices.is_cuda and self.num_epochs > 1
                 and not torch.cuda.is_current_stream_capturing())
        if ahead:
            side = _prefetch_stream(buffer.device)
            spare, pending = (pair[1] if persistent else torch.empty_like(epoch_indices)), False
        for epoch in range(self.num_epochs):
            count = self.num_mini_batches if isinstance(self.num_mini_batches, int) else self.num_mini_batches[epoch]
            if count > num_samples:
                raise ValueError(f"'num_mini_batches' ({count}) cannot exceed the number of samples ({num_samples})")
            size = num_samples // count
            if ahead:
                main = torch.cuda.current_stream()
                if pending:  # drawn during the previous epoch
                    main.wait_stream(side)
                    epoch_indices, spare = spare, epoch_indices
                    device_indices = epoch_indices
                    pending = False
                if epoch + 1 < self.num_epochs:
                    side.wait_stream(main)  # `spare` was last read by the epoch before this one
                    with torch.cuda.stream(side):
                        torch.randperm(num_samples, device=perm_device, out=spare)
                    # should the consumer abandon this generator mid-epoch (an exception, a hook stopping early), `spare`
                    # goes back to the main stream's allocator pool while the side stream may still be writing it
                    spare.record_stream(side)
                    pending = True
            elif self.shuffle and epoch > 0:
                torch.randperm(num_samples, device=perm_device, out=epoch_indices)
                if staged:
                    device_indices = epoch_indices.to(buffer.device, non_blocking=True)
            for j in range(count):
                metadata = {
                    "epoch_index": epoch,
                    "mini_batch_index": j,
                    "total_epochs": self.num_epochs,
                    "total_mini_batches": count,
                    "temporal": self.temporal,
                }
                yield metadata, device_indices[j * size : (j + 1) * size]

    def draw_epochs(self, buffer: Buffer, after: "torch.cuda.Event | None" = None, prepare: bool = True):
        """The epochs of one pass with their permutations in ONE persistent ``[E, S]`` index buffer, drawn on the draw-ahead
        stream epoch by epoch as the consumer asks for them (:class:`DrawnEpochs`; epoch 0 is drawn right here) — or None when
        a condition of :meth:`iter_indices`'s draw-ahead does not hold (no shuffle, CPU-generator permutations, ``prefetch``
        off because a hook or a dropout layer draws random numbers inside the steps).  Same generator calls in the same order
        as the epoch-by-epoch iteration.  For consumers that replay a whole epoch's minibatch steps from one hipGraph
        (template/graphs.py GraphedEpochs): each slice lives at a fixed address, an epoch may start once its event fired.
        ``after``: an event behind the last reader of the rows drawn by the previous call (the consumer's last replay); without
        it the draw waits for everything the current stream has been given so far.  ``prepare=False``: the caller refreshes
        the buffer's per-slot record itself (it draws before the fields of this update exist)."""
        if not (buffer.full and buffer.cursor == 0):
            raise RuntimeError("MiniBatchSampler requires a full buffer with cursor reset to 0")
        perm_device = self.permutation_device or buffer.device
        stock = type(self).iter_indices in (MiniBatchSampler.iter_indices,)  # a subclass that edits the iteration keeps it
        if not (stock and self.prefetch and self.shuffle and perm_device == buffer.device and perm_device.type == "cuda"
                and not torch.cuda.is_current_stream_capturing()):
            return None
        num_samples = self._get_num_samples(buffer)
        if prepare:
            buffer.prepare_sampling(self.hot_fields if self.lazy else None)
        key = (num_samples, perm_device, "epochs", self.num_epochs)
        slab = self._index_buffers.get(key)
        if slab is None:
            slab = self._index_buffers[key] = [torch.empty((self.num_epochs, num_samples), dtype=torch.int64, device=perm_device)]
        plan = []
        for epoch in range(self.num_epochs):
            count = self.num_mini_batches if isinstance(self.num_mini_batches, int) else self.num_mini_batches[epoch]
            if count > num_samples:
                raise ValueError(f"'num_mini_batches' ({count}) cannot exceed the number of samples ({num_samples})")
            size = num_samples // count
            plan.append([({"epoch_index": epoch, "mini_batch_index": j, "total_epochs": self.num_epochs,
                           "total_mini_batches": count, "temporal": self.temporal}, j * size, (j + 1) * size)
                         for j in range(count)])
        self.persistent_indices = True
        side = _prefetch_stream(perm_device)
        if after is not None:
            side.wait_event(after)  # the previous update's steps have read these rows
        else:
            side.wait_stream(torch.cuda.current_stream(perm_device))
        drawn = DrawnEpochs(slab[0], plan, side)
        drawn.draw(0)
        return drawn

    def __call__(self, buffer: Buffer):
        previous = None
        for metadata, indices in self.iter_indices(buffer):
            if previous is not None:
                previous.expire()  # its index slice is about to be reused / redrawn
            if self.lazy:
                previous = batch = buffer.gather_lazy(indices, self.temporal, self.hot_fields)
            else:
                batch = buffer.gather(indices, temporal=self.temporal)
            yield metadata, batch
        if previous is not None:
            previous.expire()

    def _get_num_samples(self, buffer: Buffer) -> int:
        return buffer.capacity * buffer.get_parallelism()


class TemporalMiniBatchSampler(MiniBatchSampler):
    """Permutes env ids and yields whole ``[T, n_envs/mb, ...]`` sequences (``:92-114``)."""

    temporal = True

    def _get_num_samples(self, buffer: Buffer) -> int:
        return buffer.get_parallelism()


class AutoMiniBatchSampler(Sampler):
    """Temporal sampling iff some top-level field name ends with ``memory`` (``:136-140``)."""

    def __init__(self, num_epochs: int = 1, num_mini_batches: int | Sequence[int] = 1, shuffle: bool = True,
                 *, permutation_device: str | torch.device | None = None, lazy: bool = True, prefetch: bool = True):
        self.num_epochs = num_epochs
        self.num_mini_batches = num_mini_batches
        self.shuffle = shuffle
        self.permutation_device = permutation_device
        self.lazy = lazy
        self.prefetch = prefetch
        self.hot_fields: set[str] = set()
        self._index_buffers: dict[tuple, list[torch.Tensor]] = {}
        self._last: MiniBatchSampler | None = None

    @property
    def persistent_indices(self) -> bool:
        return self._last is not None and self._last.persistent_indices

    def draw_epochs(self, buffer: Buffer, after=None, prepare: bool = True):
        return self._dispatch(buffer).draw_epochs(buffer, after, prepare)

    def _dispatch(self, buffer: Buffer) -> MiniBatchSampler:
        temporal = any(key.split(".")[0].endswith("memory") for key in buffer)
        cls = TemporalMiniBatchSampler if temporal else MiniBatchSampler
        sampler = cls(self.num_epochs, self.num_mini_batches, self.shuffle, permutation_device=self.permutation_device,
                      lazy=self.lazy, prefetch=self.prefetch)
        sampler.hot_fields = self.hot_fields  # the per-call sampler objects share what earlier passes learned
        sampler._index_buffers = self._index_buffers  # ... and the persistent index buffers
        self._last = sampler
        return sampler

    def __call__(self, buffer: Buffer):
        return self._dispatch(buffer)(buffer)

    def iter_indices(self, buffer: Buffer):
        return self._dispatch(buffer).iter_indices(buffer)
