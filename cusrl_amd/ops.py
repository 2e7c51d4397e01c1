"""Tensor-level entry points of the HIP hot path.

Every function takes torch tensors that already live in HBM, passes ``data_ptr()`` + sizes + torch's
current ``hipStream_t`` through the C ABI (``include/cusrl_hip.h``) and returns torch tensors it allocated
with torch's caching allocator.  PyTorch is plumbing here (memory, streams); the arithmetic is in
``cusrl_amd/csrc/*.hip``.  Non-device tensors are rejected — there is deliberately no CPU path.
"""

from __future__ import annotations

import os
from collections.abc import Sequence

import torch

from cusrl_amd import _native
from cusrl_amd._native import Field, PackedField, check

__all__ = [
    "DeferredLoss",
    "LaunchObserver",
    "RecordPack",
    "adv_stats_finalize",
    "amp_prepare",
    "amp_prepare_supported",
    "amp_style_reward_",
    "amp_style_reward_mean_",
    "mse_loss_fwd_bwd",
    "sumsq_fwd_bwd",
    "bce_pair_fwd_bwd",
    "reward_shaping_",
    "buffer_push",
    "col_stats",
    "compact_flags",
    "episode_stats",
    "gae",
    "gather_rows",
    "gather_rows_packed",
    "merge_mean_var",
    "next_value",
    "normal_sample_logp",
    "categorical_sample_logp",
    "gru_gates_forward",
    "input_layer_backward",
    "input_layer_supported",
    "gru_gates_backward",
    "lstm_gates_forward",
    "lstm_gates_backward",
    "rnn_cell_forward",
    "rnn_cell_backward",
    "normalize_",
    "policy_terms_fwd",
    "policy_terms_bwd",
    "categorical_terms_fwd",
    "categorical_terms_bwd",
    "normalize_from_partials_",
    "ppo_loss_categorical_fwd_bwd",
    "ppo_loss_fwd_bwd",
    "value_loss_fwd_bwd",
    "masked_col_stats",
    "relu_backward_bias",
    "require_device",
    "rms_merge_",
    "rnd_reward_",
    "rms_normalize",
    "scatter_rows",
    "set_launch_observer",
    "splice_rows",
    "step_epilogue",
    "window_indices",
]


class LaunchObserver:
    """Optional HIP-event bracketing of individual launches, placed directly around the C call (after all host-side
    preparation), on the stream the kernel is launched on.  Used by ``bench.py``; ``None`` in normal operation."""

    def __init__(self, only: set[str] | None = None):
        # timing events are not free on this stack (~20-40 us of stream bubble per pair): observe few launches
        self.only = only
        self.records: dict[str, list[tuple[torch.cuda.Event, torch.cuda.Event, int]]] = {}
        self._open: torch.cuda.Event | None = None

    def begin(self):
        self._open = torch.cuda.Event(enable_timing=True)
        self._open.record()

    def end(self, name: str, nbytes: int):
        event = torch.cuda.Event(enable_timing=True)
        event.record()
        self.records.setdefault(name, []).append((self._open, event, nbytes))


_observer: LaunchObserver | None = None


def set_launch_observer(observer: LaunchObserver | None) -> None:
    global _observer
    _observer = observer


def _observed(name: str, nbytes_fn, call) -> None:
    """Run ``call()`` (a single C-ABI launch returning its status) with optional event bracketing."""
    observer = _observer
    if observer is None or (observer.only is not None and name not in observer.only) or torch.cuda.is_current_stream_capturing():
        check(call(), name)
        return
    observer.begin()
    status = call()
    observer.end(name, nbytes_fn())
    check(status, name)


def require_device(tensor: torch.Tensor, name: str = "tensor") -> torch.Tensor:
    if not tensor.is_cuda:
        raise RuntimeError(
            f"cusrl_amd: '{name}' lives on {tensor.device}; the rollout + PPO-update hot path only runs as HIP "
            "kernels on an MI355X device tensor (no CPU fallback exists by design)"
        )
    return tensor


def _f32(tensor: torch.Tensor, name: str) -> torch.Tensor:
    require_device(tensor, name)
    if tensor.dtype != torch.float32:
        raise TypeError(f"'{name}' must be float32, got {tensor.dtype}")
    return tensor if tensor.is_contiguous() else tensor.contiguous()


def _flag(tensor: torch.Tensor, name: str) -> torch.Tensor:
    require_device(tensor, name)
    if tensor.dtype not in (torch.bool, torch.uint8):
        raise TypeError(f"'{name}' must have dtype bool, got {tensor.dtype}")
    return tensor if tensor.is_contiguous() else tensor.contiguous()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """Handle of torch's current stream on the current device (every launch of this module goes there).  The raw
    lookup skips the ``torch.cuda.Stream`` object the public accessor builds — ~2 us per launch on the host, and the
    rollout loop is host-bound."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _row_bytes(tensor: torch.Tensor, lead_dims: int) -> int:
    n = tensor.element_size()
    for s in tensor.shape[lead_dims:]:
        n *= s
    return n


# ------------------------------------------------------------------------------------------------ a1
def buffer_push(pairs: Sequence[tuple[torch.Tensor, torch.Tensor]], cursor: int, parallelism: int) -> None:
    """``storage[cursor] = step`` for every ``(step [N,...], storage [T,N,...])`` pair in one launch.

    Replaces the per-leaf index_put chain of cusrl/template/buffer.py:134-146.
    """
    lib = _native.lib()
    stream = _stream()
    for start in range(0, len(pairs), _native.MAX_FIELDS):
        chunk = pairs[start : start + _native.MAX_FIELDS]
        table = (Field * len(chunk))()
        for i, (step, storage) in enumerate(chunk):
            table[i].src = step.data_ptr()
            table[i].dst = storage.data_ptr()
            table[i].row_bytes = _row_bytes(step, 1)
        _observed(
            "cusrl_buffer_push",
            lambda: sum(2 * step.numel() * step.element_size() for step, _ in chunk),
            lambda: lib.cusrl_buffer_push(table, len(chunk), cursor, parallelism, stream),
        )


def make_push_table(leaves: Sequence[tuple[torch.Tensor, tuple]]):
    """Pre-filled ``cusrl_field_t`` array for a fixed set of ``(storage [T, N, ...], step shape)`` leaves: destination
    pointers and row sizes never change between pushes, only the source pointers do."""
    table = (Field * max(len(leaves), 1))()
    for i, (storage, shape) in enumerate(leaves):
        table[i].dst = storage.data_ptr()
        row = storage.element_size()
        for s in shape[1:]:
            row *= s
        table[i].row_bytes = row
    return table


def push_table(table, count: int, cursor: int, parallelism: int, through: tuple | None = None) -> None:
    """Launch ``cusrl_buffer_push`` on a table whose ``src`` pointers were just filled.  ``through`` =
    ``(record tensor, record_bytes, int32 offsets array)``: the leaves with an offset >= 0 are also written into the
    per-slot record from the same registers (``cusrl_buffer_push_through``)."""
    lib = _native.lib()
    stream = _stream()
    if through is None:
        _observed(
            "cusrl_buffer_push",
            lambda: sum(2 * parallelism * table[i].row_bytes for i in range(count)),
            lambda: lib.cusrl_buffer_push(table, count, cursor, parallelism, stream),
        )
        return
    record, record_bytes, offsets = through
    _observed(
        "cusrl_buffer_push_through",
        lambda: sum((2 + (offsets[i] >= 0)) * parallelism * table[i].row_bytes for i in range(count)),
        lambda: lib.cusrl_buffer_push_through(table, count, cursor, parallelism, record.data_ptr(), record_bytes, offsets, stream),
    )


# ------------------------------------------------------------------------------------------------ a7 / a8
def _row_elems(storage: torch.Tensor) -> int:
    n = 1
    for d in storage.shape[2:]:
        n *= d
    return n


def gather_rows(
    storages: Sequence[torch.Tensor],
    indices: torch.Tensor,
    capacity: int,
    parallelism: int,
    temporal: bool = False,
    out: Sequence[torch.Tensor] | None = None,
) -> list[torch.Tensor]:
    """Minibatch gather of every leaf in one launch.

    ``storage.flatten(0, 1)[indices]`` (cusrl/sampler/mini_batch_sampler.py:87-89) or ``storage[:, indices]``
    (``:113-114``) for each ``[T, N, ...]`` leaf.  ``out``: contiguous destinations the caller owns (e.g. two halves of
    one joint batch), one per leaf.
    """
    require_device(indices, "indices")
    if indices.dtype != torch.int64:
        raise TypeError(f"'indices' must be int64, got {indices.dtype}")
    if not indices.is_contiguous():
        indices = indices.contiguous()
    batch = indices.numel()
    lead = (capacity, batch) if temporal else (batch,)
    if out is None:
        outputs = [torch.empty(lead + tuple(s.shape[2:]), dtype=s.dtype, device=s.device) for s in storages]
    else:
        outputs = list(out)
        for dst, src in zip(outputs, storages):
            if not dst.is_contiguous() or dst.dtype != src.dtype or dst.numel() != batch * (capacity if temporal else 1) * _row_elems(src):
                raise ValueError("gather_rows: an 'out' tensor does not match its leaf (contiguous, same dtype, batch rows)")
    if batch == 0:
        return outputs
    lib = _native.lib()
    stream = _stream()
    for start in range(0, len(storages), _native.MAX_FIELDS):
        chunk = range(start, min(start + _native.MAX_FIELDS, len(storages)))
        table = (Field * len(chunk))()
        for i, k in enumerate(chunk):
            src = storages[k]
            if not src.is_contiguous():
                raise ValueError("buffer leaves must be contiguous [capacity, parallelism, ...] tensors")
            table[i].src = src.data_ptr()
            table[i].dst = outputs[k].data_ptr()
            table[i].row_bytes = _row_bytes(src, 2)
        rows = batch * (capacity if temporal else 1)
        _observed(
            "cusrl_gather_rows",
            lambda: rows * sum(2 * _row_bytes(storages[k], 2) for k in chunk) + batch * 8,
            lambda: lib.cusrl_gather_rows(table, len(chunk), indices.data_ptr(), batch, capacity, parallelism, int(temporal), stream),
        )
    return outputs


def window_indices(start: torch.Tensor, env: torch.Tensor, length: int, capacity: int, parallelism: int,
                   cursor: int | None) -> torch.Tensor:
    """Flat slots ``[length * B]`` of ``B`` temporal windows (``data[time_indices, env_indices]`` of
    cusrl/sampler/random_sampler.py:101-113 as one index list); ``cursor`` = oldest row of a FULL ring, else None."""
    require_device(start, "start_indices"), require_device(env, "env_indices")
    if start.dtype != torch.int64 or env.dtype != torch.int64 or start.shape != env.shape:
        raise TypeError("window_indices: int64 index vectors of equal length are required")
    start, env = start.contiguous(), env.contiguous()
    out = torch.empty(length * start.numel(), dtype=torch.int64, device=start.device)
    check(_native.lib().cusrl_window_indices(start.data_ptr(), env.data_ptr(), out.data_ptr(), start.numel(), length,
                                             capacity, parallelism, 0 if cursor is None else cursor, _stream()),
          "cusrl_window_indices")
    return out


class RecordPack:
    """Leaves of a rollout buffer interleaved into ONE record per slot, so that a randomly sampled slot costs as few
    128-byte memory lines as possible (MI355X fetches a whole line for a random row of any size <= 128 B, measured:
    profiles/r02/pmc_summary.json).  Two uses:

    * the narrow leaves (1-8 bytes per slot: log-prob, value, reward, next_value, advantage, return, flags): 27 B of the
      ``ppo`` buffer -> one 32-byte record instead of nine separate line fetches;
    * the HOT set — every leaf one training step reads, wide ones included (observation 192 B + action 48 B + log-prob,
      advantage, return 12 B + done 1 B = 253 B -> a 256-byte record = exactly two lines per sampled slot).

    ``build()`` (re)writes the record from the leaves — once per update, after the ``pre_update`` hooks have produced
    their fields; :func:`gather_rows_packed` then reads the record instead of the leaves.  Layout: wide leaves (a
    multiple of 16 bytes) first at 16-byte offsets, then 8/4-byte, 2-byte and 1-byte entries; the record size is the next
    of 16 / 32 / 64 / a multiple of 128 bytes."""

    NARROW = (1, 2, 4, 8)
    MAX_BYTES = 1024

    @staticmethod
    def eligible(storage: torch.Tensor, wide: bool = False) -> bool:
        if not (storage.is_cuda and storage.is_contiguous() and storage.dim() >= 2):
            return False
        width = _row_bytes(storage, 2)
        return width in RecordPack.NARROW or (wide and width % 16 == 0 and 0 < width <= 512 and storage.data_ptr() % 16 == 0)

    @staticmethod
    def _entries(width: int) -> int:
        return 0 if width >= 16 else (2 if width == 8 else 1)

    @staticmethod
    def _size(total: int) -> int:
        return next((size for size in (16, 32, 64) if size >= total), -(-total // 128) * 128)

    @classmethod
    def plan(cls, storages: dict[str, torch.Tensor], hot: Sequence[str] | None = None) -> list[str]:
        """Leaves to pack.  ``hot`` (leaf names) given: exactly those, wide ones included, when they are all eligible and
        fit; otherwise the narrow leaves in storage order while they fit 64 bytes / ``MAX_PACKED`` entries (packing a
        single leaf would only add a copy, so fewer than two means no record at all)."""
        if hot:
            chosen = [name for name in storages if name in hot]
            widths = [_row_bytes(storages[name], 2) for name in chosen]
            if (len(chosen) >= 2 and all(cls.eligible(storages[name], wide=True) for name in chosen)
                    and sum(cls._entries(w) for w in widths) <= _native.MAX_PACKED
                    and sum(1 for w in widths if w >= 16) <= _native.MAX_FIELDS and cls._size(sum(widths)) <= cls.MAX_BYTES):
                return chosen
        chosen, total, entries = [], 0, 0
        for name, storage in storages.items():
            if not cls.eligible(storage):
                continue
            width = _row_bytes(storage, 2)
            slots = cls._entries(width)
            if entries + slots > _native.MAX_PACKED or total + width > 64:
                break
            chosen.append(name)
            total += width
            entries += slots
        return chosen if len(chosen) >= 2 else []

    def __init__(self, storages: dict[str, torch.Tensor]):
        # widest first: wide leaves land on 16-byte offsets, every narrow entry on a multiple of its width
        names = sorted(storages, key=lambda k: -_row_bytes(storages[k], 2))
        self.leaves = {name: storages[name] for name in names}
        first = next(iter(self.leaves.values()))
        self.rows = first.shape[0] * first.shape[1]
        self.offsets: dict[str, int] = {}
        offset = 0
        for name, storage in self.leaves.items():
            if not self.eligible(storage, wide=True) or storage.shape[0] * storage.shape[1] != self.rows:
                raise ValueError(f"leaf '{name}' cannot be packed")
            self.offsets[name] = offset
            offset += _row_bytes(storage, 2)
        self.used_bytes = offset
        self.record_bytes = self._size(offset)
        widths = [_row_bytes(t, 2) for t in self.leaves.values()]
        if self.record_bytes > self.MAX_BYTES or sum(self._entries(w) for w in widths) > _native.MAX_PACKED:
            raise ValueError("the packed leaves exceed one record (1024 bytes / 16 narrow entries)")
        self.record = torch.empty((self.rows, self.record_bytes), dtype=torch.uint8, device=first.device)
        self.key = tuple((name, t.data_ptr(), _row_bytes(t, 2)) for name, t in self.leaves.items())
        self._table = (PackedField * len(self.leaves))()
        for slot, (name, storage) in zip(self._table, self.leaves.items()):
            slot.ptr, slot.offset, slot.width = storage.data_ptr(), self.offsets[name], _row_bytes(storage, 2)

    def build(self, names: Sequence[str] | None = None) -> None:
        """(Re)write the record from the leaves — all of them, or only ``names`` (the leaves that changed since the
        record last held them: the other bytes of every record stay as they are)."""
        if names is None:
            table, count, moved = self._table, len(self.leaves), self.used_bytes
        else:
            chosen = [name for name in self.leaves if name in set(names)]
            if not chosen:
                return
            table = (PackedField * len(chosen))()
            for slot, name in zip(table, chosen):
                storage = self.leaves[name]
                slot.ptr, slot.offset, slot.width = storage.data_ptr(), self.offsets[name], _row_bytes(storage, 2)
            count, moved = len(chosen), sum(_row_bytes(self.leaves[name], 2) for name in chosen)
        owned = self._owned_chunks(None if names is None else set(names))
        if owned is not None:
            _observed(
                "cusrl_pack_rows_owned",
                lambda: self.rows * 2 * moved,
                lambda: _native.lib().cusrl_pack_rows_owned(table, count, self.record.data_ptr(), self.record_bytes, self.rows,
                                                            owned[0], owned[1], _stream()),
            )
            return
        _observed(
            "cusrl_pack_rows",
            lambda: self.rows * 2 * moved,
            lambda: _native.lib().cusrl_pack_rows(table, count, self.record.data_ptr(), self.record_bytes, self.rows, _stream()),
        )

    def _owned_chunks(self, names: set[str] | None) -> tuple[int, int] | None:
        """``(first chunk, chunks)`` when the narrow leaves of this record occupy at most two 16-byte chunks of their own
        (the layout puts them behind the wide leaves, which end on a chunk boundary) and the call writes ALL of them: the
        kernel may then store those chunks whole.  A partial repack must leave the other narrow leaves' bytes alone."""
        narrow = [name for name, storage in self.leaves.items() if _row_bytes(storage, 2) < 16]
        if not narrow or (names is not None and not all(name in names for name in narrow)):
            return None
        start = min(self.offsets[name] for name in narrow)
        if start % 16:
            return None
        chunks = -(-(self.used_bytes - start) // 16)
        return (start // 16, chunks) if chunks <= 2 else None

    def through_offsets(self, leaves: Sequence[str]):
        """int32 array for ``cusrl_buffer_push_through``: the record offset of every pushed leaf that can be written
        through (a wide leaf of this record: whole 16-byte chunks), -1 for the others; None when there is none."""
        import ctypes

        offsets = (ctypes.c_int32 * max(len(leaves), 1))()
        any_through = False
        for i, name in enumerate(leaves):
            storage = self.leaves.get(name)
            wide = (storage is not None and _row_bytes(storage, 2) % 16 == 0 and storage.data_ptr() % 16 == 0
                    and storage.shape[1] * _row_bytes(storage, 2) < 2**32)
            offsets[i] = self.offsets[name] if wide else -1
            any_through |= wide
        return offsets if any_through else None


def gather_rows_packed(
    storages: Sequence[torch.Tensor],
    pack: RecordPack | None,
    packed_names: Sequence[str],
    indices: torch.Tensor,
    capacity: int,
    parallelism: int,
    temporal: bool = False,
    out: Sequence[torch.Tensor] | None = None,
    packed_out: Sequence[torch.Tensor] | None = None,
) -> tuple[list[torch.Tensor], list[torch.Tensor]]:
    """:func:`gather_rows` for ``storages`` plus, from the SAME launch, the leaves ``packed_names`` of ``pack`` read
    through its per-slot record (one sector per sampled slot for all of them).  Results are identical to gathering
    the leaves themselves as long as the record is current (``pack.build()`` after the last write to a packed leaf).
    ``out`` / ``packed_out``: destinations the caller owns, one per plain / packed leaf."""
    if not packed_names:
        return gather_rows(storages, indices, capacity, parallelism, temporal, out=out), []
    if len(storages) > _native.MAX_FIELDS:
        raise ValueError("gather_rows_packed: too many plain leaves for one launch")
    require_device(indices, "indices")
    if indices.dtype != torch.int64:
        raise TypeError(f"'indices' must be int64, got {indices.dtype}")
    if not indices.is_contiguous():
        indices = indices.contiguous()
    batch = indices.numel()
    lead = (capacity, batch) if temporal else (batch,)
    sources = [pack.leaves[name] for name in packed_names]
    rows_out = batch * (capacity if temporal else 1)

    def destinations(given, leaves):
        if given is None:
            return [torch.empty(lead + tuple(s.shape[2:]), dtype=s.dtype, device=s.device) for s in leaves]
        given = list(given)
        for dst, src in zip(given, leaves):
            if not dst.is_contiguous() or dst.dtype != src.dtype or dst.numel() != rows_out * _row_elems(src):
                raise ValueError("gather_rows_packed: an 'out' tensor does not match its leaf (contiguous, same dtype, batch rows)")
        return given

    outputs, packed_outputs = destinations(out, storages), destinations(packed_out, sources)
    if batch == 0:
        return outputs, packed_outputs
    table = (Field * max(len(storages), 1))()
    for i, src in enumerate(storages):
        if not src.is_contiguous():
            raise ValueError("buffer leaves must be contiguous [capacity, parallelism, ...] tensors")
        table[i].src, table[i].dst, table[i].row_bytes = src.data_ptr(), outputs[i].data_ptr(), _row_bytes(src, 2)
    packed_table = (PackedField * len(packed_names))()
    for slot, name, out in zip(packed_table, packed_names, packed_outputs):
        slot.ptr, slot.offset, slot.width = out.data_ptr(), pack.offsets[name], _row_bytes(pack.leaves[name], 2)
    rows = batch * (capacity if temporal else 1)
    lib = _native.lib()
    stream = _stream()
    _observed(
        "cusrl_gather_rows_packed",
        lambda: rows * (sum(2 * _row_bytes(s, 2) for s in storages) + sum(2 * _row_bytes(s, 2) for s in sources)) + batch * 8,
        lambda: lib.cusrl_gather_rows_packed(table, len(storages), pack.record.data_ptr(), pack.record_bytes, packed_table,
                                             len(packed_names), indices.data_ptr(), batch, capacity, parallelism, int(temporal), stream),
    )
    return outputs, packed_outputs


# ------------------------------------------------------------------------------------------------ a3
def next_value(
    value: torch.Tensor,
    terminated: torch.Tensor,
    truncated: torch.Tensor,
    last_value: torch.Tensor,
    termination_value: float,
    truncated_uses_own_value: bool,
    out: torch.Tensor,
) -> torch.Tensor:
    """Bootstrap target of cusrl/hook/on_policy/value.py:66-70,79-80; returns the per-block truncated counters."""
    value = _f32(value, "value")
    T, N, D = value.shape
    terminated, truncated = _flag(terminated, "terminated"), _flag(truncated, "truncated")
    last_value = _f32(last_value, "last_value")
    if last_value.numel() != N * D or out.shape != value.shape or not out.is_contiguous():
        raise ValueError("next_value: inconsistent shapes")
    lib = _native.lib()
    block_counts = torch.empty(max(int(lib.cusrl_flag_blocks(T * N)), 1), dtype=torch.int32, device=value.device)
    out_ptr = _f32(out, "next_value").data_ptr()
    _observed(
        "cusrl_next_value",
        lambda: T * N * (8 * D + 2),
        lambda: lib.cusrl_next_value(
            value.data_ptr(), terminated.data_ptr(), truncated.data_ptr(), last_value.data_ptr(),
            float(termination_value), int(truncated_uses_own_value), out_ptr, block_counts.data_ptr(), T, N, D, _stream(),
        ),
    )
    return block_counts


def compact_flags(flags: torch.Tensor, block_counts: torch.Tensor | None = None,
                  count_out: torch.Tensor | None = None, *, scratch: dict | None = None) -> tuple[torch.Tensor, torch.Tensor]:
    """Ascending flat slots whose flag is set + their number (int32[1]); no host synchronisation.  ``count_out``
    may be a pinned host tensor (the kernel stores the count there with system scope; see :class:`HostCounter`).
    ``scratch`` (a dict the caller keeps) lets a per-step caller reuse the counter and index buffers: the returned
    indices are then only valid until the next call with the same scratch."""
    flags = _flag(flags, "flags")
    n = flags.numel()
    lib = _native.lib()
    recount = block_counts is None
    if scratch is not None and scratch.get("n") == n and scratch.get("device") == flags.device:
        indices = scratch["indices"]
        if recount:
            block_counts = scratch["counts"]
    else:
        if recount:
            block_counts = torch.empty(max(int(lib.cusrl_flag_blocks(n)), 1), dtype=torch.int32, device=flags.device)
        indices = torch.empty(n, dtype=torch.int64, device=flags.device)
        if scratch is not None and recount:
            scratch.update(n=n, device=flags.device, indices=indices, counts=block_counts)
    if count_out is not None:
        if count_out.dtype != torch.int32 or count_out.numel() != 1 or not (count_out.is_cuda or count_out.is_pinned()):
            raise TypeError("'count_out' must be a 1-element int32 tensor on the device or in pinned host memory")
        count = count_out
    else:
        count = torch.empty(1, dtype=torch.int32, device=flags.device)
    check(
        lib.cusrl_compact_flags(flags.data_ptr(), n, block_counts.data_ptr(), int(recount), indices.data_ptr(), count.data_ptr(), _stream()),
        "cusrl_compact_flags",
    )
    return indices, count


def assign_rows(dst: torch.Tensor, indices: torch.Tensor, src: torch.Tensor) -> None:
    """``dst[indices] = src`` for a contiguous ``dst [N, ...]`` and ``src [K, ...]`` of the same dtype (the reset
    observations spliced into the rollout's current observation, environment.py:365-379) — the 16-byte-lane row scatter
    instead of torch's general ``index_put_``."""
    require_device(src, "src"), require_device(dst, "dst"), require_device(indices, "indices")
    if src.dtype != dst.dtype or indices.dtype != torch.int64 or not dst.is_contiguous() or src.shape[1:] != dst.shape[1:]:
        raise TypeError("assign_rows: dtype/layout mismatch")
    K = indices.numel()
    if K == 0:
        return
    if src.shape[0] != K:
        raise ValueError("assign_rows: one source row per index is required")
    src, indices = src.contiguous(), indices.contiguous()
    check(
        _native.lib().cusrl_scatter_rows(src.data_ptr(), indices.data_ptr(), dst.data_ptr(), K, _row_bytes(src, 1), None, _stream()),
        "cusrl_scatter_rows",
    )
    _modified_in_place(dst)


def splice_rows(src: torch.Tensor, init: torch.Tensor, indices: torch.Tensor, count: torch.Tensor, done: torch.Tensor,
                dst: torch.Tensor) -> torch.Tensor:
    """``dst = src`` with the reset rows spliced in (``update_observation_and_state``, environment.py:365-379, fused with
    the copy into the next act step's input): ``dst[n] = src[n]`` where ``done[n]`` is clear, ``dst[indices[k]] = init[k]``
    for ``k < count`` (read on the device).  ``done`` and ``(indices, count)`` must come from the same step epilogue."""
    for tensor, name in ((src, "src"), (init, "init"), (indices, "indices"), (count, "count"), (done, "done"), (dst, "dst")):
        require_device(tensor, name)
    N = src.shape[0]
    if (src.dtype != dst.dtype or init.dtype != dst.dtype or src.shape != dst.shape or init.shape != dst.shape
            or not (src.is_contiguous() and init.is_contiguous() and dst.is_contiguous())):
        raise TypeError("splice_rows: src, init and dst must be contiguous tensors of one shape and dtype")
    if indices.dtype != torch.int64 or indices.numel() < N or count.dtype != torch.int32 or count.numel() != 1 or done.numel() != N:
        raise TypeError("splice_rows: indices int64[>= N], count int32[1], done [N] flags are required")
    if dst.data_ptr() in (src.data_ptr(), init.data_ptr()):
        raise ValueError("splice_rows: dst must not alias src or init")
    done = _flag(done, "done")
    check(_native.lib().cusrl_splice_rows(src.data_ptr(), init.data_ptr(), indices.data_ptr(), count.data_ptr(), done.data_ptr(),
                                          dst.data_ptr(), N, _row_bytes(src, 1), _stream()), "cusrl_splice_rows")
    _modified_in_place(dst)
    return dst


class HostCounter:
    """A pinned, device-mapped int32 the host polls for a kernel's result.

    ``int(count.item())`` on a device scalar is a device->host copy plus a stream synchronisation — ~30 us on this
    stack even when the GPU is already idle.  A kernel that stores its scalar result straight into pinned host memory
    (system-scope store) lets the host spin on it and continue a few microseconds after the kernel retires."""

    def __init__(self):
        self.tensor = torch.empty(1, dtype=torch.int32).pin_memory()
        self._view = self.tensor.numpy()

    def arm(self) -> torch.Tensor:
        self._view[0] = -1
        return self.tensor

    def wait(self, timeout: float = 0.01) -> int:
        import time

        view, deadline = self._view, None
        while True:
            value = int(view[0])
            if value >= 0:
                return value
            if deadline is None:
                deadline = time.perf_counter() + timeout
            elif time.perf_counter() > deadline:  # something is slow (first launch, profiler): block the usual way
                torch.cuda.current_stream().synchronize()
                value = int(view[0])
                if value < 0:
                    raise RuntimeError("HostCounter: the kernel retired without publishing its count")
                return value


def scatter_rows(src: torch.Tensor, indices: torch.Tensor, dst: torch.Tensor, count: torch.Tensor | None = None) -> None:
    """``dst.flatten(0, 1)[indices] = src`` (cusrl/hook/on_policy/value.py:78).  With ``count`` (a 1-element int32 on
    the device or in pinned host memory) only the first ``min(len(indices), count)`` rows are written — the number is
    read by the kernel, so a fixed-capacity launch can follow an on-device compaction without a host read."""
    require_device(src, "src"), require_device(dst, "dst"), require_device(indices, "indices")
    if count is not None and (count.dtype != torch.int32 or count.numel() != 1 or not (count.is_cuda or count.is_pinned())):
        raise TypeError("'count' must be a 1-element int32 tensor on the device or in pinned host memory")
    if src.dtype != dst.dtype or indices.dtype != torch.int64 or not dst.is_contiguous():
        raise TypeError("scatter_rows: dtype/layout mismatch")
    src, indices = src.contiguous(), indices.contiguous()
    K = indices.numel()
    if K == 0:
        return
    check(
        _native.lib().cusrl_scatter_rows(src.data_ptr(), indices.data_ptr(), dst.data_ptr(), K, _row_bytes(src, 1),
                                         None if count is None else count.data_ptr(), _stream()),
        "cusrl_scatter_rows",
    )
    _modified_in_place(dst)


# ------------------------------------------------------------------------------------------------ a4
def gae(
    reward: torch.Tensor,
    value: torch.Tensor,
    next_value_: torch.Tensor,
    done: torch.Tensor,
    gamma: float,
    lamda: float,
    lamda_value: float | None,
    advantage: torch.Tensor | None = None,
    ret: torch.Tensor | None = None,
    with_stats: bool = True,
) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor | None]:
    """Fused delta + backward scan + return (+ advantage {sum, sumsq} partials) — gae.py:8-20, 85-110."""
    reward, value, next_value_ = _f32(reward, "reward"), _f32(value, "value"), _f32(next_value_, "next_value")
    done = _flag(done, "done")
    if reward.dim() < 2 or reward.shape != value.shape or reward.shape != next_value_.shape:
        raise ValueError(f"gae: reward/value/next_value shapes differ: {reward.shape}, {value.shape}, {next_value_.shape}")
    T, N = reward.shape[:2]
    D = reward.numel() // max(T * N, 1)
    if done.numel() != T * N:
        raise ValueError(f"gae: 'done' must be [T, N, 1]; got {tuple(done.shape)}")
    for out, name in ((advantage, "advantage"), (ret, "return")):
        if out is not None and (not out.is_contiguous() or out.shape != reward.shape):
            raise ValueError(f"gae: the '{name}' output must be a contiguous tensor of the reward's shape (results are "
                             "written in place; a strided view would silently receive nothing)")
    advantage = torch.empty_like(reward) if advantage is None else _f32(advantage, "advantage")
    ret = torch.empty_like(reward) if ret is None else _f32(ret, "return")
    lib = _native.lib()
    partials = None
    if with_stats:
        partials = torch.empty((max(int(lib.cusrl_gae_num_partials(T, N, D)), 1), D, 2), dtype=torch.float64, device=reward.device)
    _observed(
        "cusrl_gae",
        lambda: T * N * (20 * D + 1),
        lambda: lib.cusrl_gae(
            reward.data_ptr(), value.data_ptr(), next_value_.data_ptr(), done.data_ptr(), advantage.data_ptr(),
            ret.data_ptr(), None if partials is None else partials.data_ptr(), T, N, D, float(gamma), float(lamda),
            -1.0 if lamda_value is None else float(lamda_value), _stream(),
        ),
    )
    return advantage, ret, partials


# ------------------------------------------------------------------------------------------------ a5 / a6
def col_stats(x: torch.Tensor) -> torch.Tensor:
    """Per-channel {sum, sumsq} partials of ``x [..., D]`` (first pass of advantage.py:111)."""
    x = _f32(x, "x")
    D = x.shape[-1]
    rows = x.numel() // max(D, 1)
    lib = _native.lib()
    partials = torch.empty((max(int(lib.cusrl_col_stats_num_partials(rows, D)), 1), D, 2), dtype=torch.float64, device=x.device)
    check(lib.cusrl_col_stats(x.data_ptr(), rows, D, partials.data_ptr(), _stream()), "cusrl_col_stats")
    return partials


def adv_stats_finalize(partials: torch.Tensor, count: int) -> tuple[torch.Tensor, torch.Tensor]:
    """``var, mean`` (unbiased) from block partials, fixed summation order.  The two are the halves of ONE ``[2 D]`` row
    (``mean | var`` — what a cross-rank merge gathers: ``packed_mean_var`` hands it over without a ``torch.cat``)."""
    P, D, _ = partials.shape
    row = torch.empty(2 * D, dtype=torch.float32, device=partials.device)
    mean, var = row[:D], row[D:]
    check(
        _native.lib().cusrl_stats_finalize(partials.data_ptr(), P, D, count, mean.data_ptr(), var.data_ptr(), _stream()),
        "cusrl_stats_finalize",
    )
    return var, mean


def _modified_in_place(tensor: torch.Tensor) -> None:
    """Tell torch that a raw kernel wrote into ``tensor``: bumps the version counter every alias shares — what
    ``Buffer`` compares to know whether the per-slot record still mirrors a leaf (and what autograd checks)."""
    torch.autograd.graph.increment_version(tensor)


def normalize_(x: torch.Tensor, mean: torch.Tensor, var: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    """``x.sub_(mean).div_((var + eps).sqrt())`` in place (advantage.py:114-115)."""
    require_device(x, "x")
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise TypeError("normalize_: expected a contiguous float32 tensor")
    D = x.shape[-1]
    mean, var = _f32(mean, "mean"), _f32(var, "var")
    _observed(
        "cusrl_normalize",
        lambda: x.numel() * 8,
        lambda: _native.lib().cusrl_normalize(x.data_ptr(), mean.data_ptr(), var.data_ptr(), eps, x.numel() // max(D, 1), D, _stream()),
    )
    _modified_in_place(x)
    return x


def normalize_from_partials_(x: torch.Tensor, partials: torch.Tensor, count: int, eps: float = 1e-8) -> tuple[torch.Tensor, torch.Tensor]:
    """:func:`adv_stats_finalize` + :func:`normalize_` as ONE launch (single-process case); returns ``(var, mean)``."""
    require_device(x, "x")
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise TypeError("normalize_from_partials_: expected a contiguous float32 tensor")
    P, D, _ = partials.shape
    if x.shape[-1] != D:
        raise ValueError("normalize_from_partials_: the partials belong to another tensor")
    mean = torch.empty(D, dtype=torch.float32, device=x.device)
    var = torch.empty(D, dtype=torch.float32, device=x.device)
    _observed(
        "cusrl_normalize_from_partials",
        lambda: x.numel() * 8,
        lambda: _native.lib().cusrl_normalize_from_partials(x.data_ptr(), partials.data_ptr(), P, count, eps, x.numel() // max(D, 1), D,
                                                            mean.data_ptr(), var.data_ptr(), _stream()),
    )
    _modified_in_place(x)
    return var, mean


def packed_mean_var(mean: torch.Tensor, var: torch.Tensor) -> torch.Tensor:
    """``[mean | var]`` as one contiguous row: the halves of :func:`adv_stats_finalize`'s row as they are (no launch), else a
    ``torch.cat``."""
    D = mean.numel()
    if (mean.dim() == var.dim() == 1 and var.numel() == D and mean.is_contiguous() and var.is_contiguous()
            and mean.untyped_storage().data_ptr() == var.untyped_storage().data_ptr()
            and var.storage_offset() == mean.storage_offset() + D):
        return mean.as_strided((2 * D,), (1,), mean.storage_offset())
    return torch.cat((mean, var), dim=0)


def normalize_from_gathered_(x: torch.Tensor, gathered: torch.Tensor, eps: float = 1e-8) -> tuple[torch.Tensor, torch.Tensor]:
    """:func:`merge_mean_var` + :func:`normalize_` as ONE launch: ``gathered [W, 2 D]`` holds every rank's ``mean | var``;
    returns the merged ``(var, mean)``."""
    require_device(x, "x")
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise TypeError("normalize_from_gathered_: expected a contiguous float32 tensor")
    gathered = _f32(gathered, "gathered")
    D = x.shape[-1]
    if gathered.dim() != 2 or gathered.shape[1] != 2 * D:
        raise ValueError("normalize_from_gathered_: expected [W, 2 D] rows of mean | var")
    mean = torch.empty(D, dtype=torch.float32, device=x.device)
    var = torch.empty(D, dtype=torch.float32, device=x.device)
    _observed(
        "cusrl_normalize_from_gathered",
        lambda: x.numel() * 8,
        lambda: _native.lib().cusrl_normalize_from_gathered(x.data_ptr(), gathered.data_ptr(), gathered.shape[0], eps,
                                                            x.numel() // max(D, 1), D, mean.data_ptr(), var.data_ptr(), _stream()),
    )
    _modified_in_place(x)
    return var, mean


def merge_mean_var(gathered: torch.Tensor, mean: torch.Tensor, var: torch.Tensor) -> None:
    """Equal-weight cross-rank merge of distributed.py:175-183 from the all-gathered ``[W, 2D]`` rows."""
    gathered = _f32(gathered, "gathered")
    W, twoD = gathered.shape
    check(
        _native.lib().cusrl_merge_mean_var(gathered.data_ptr(), W, twoD // 2, _f32(mean, "mean").data_ptr(), _f32(var, "var").data_ptr(), _stream()),
        "cusrl_merge_mean_var",
    )


# ------------------------------------------------------------------------------------------------ a9 - a13
LOSS_DEFER = 1  # CUSRL_LOSS_DEFER


class DeferredLoss:
    """Running sums of the objective's five block partials over the replays of ONE captured minibatch step
    (``CUSRL_LOSS_DEFER``): nothing inside an optimizer step reads the loss VALUES — the backward takes a unit gradient —
    so the captured step skips the one-block finalize launch; every block adds its sums to its own row of ``rows``
    (persistent, zero-filled here, outside the capture) and :meth:`drain` forms the per-replay means once per update on
    the host.  ``weights`` = (w_val, w_sur, w_ent) the capture froze.  The value term evaluated by its own launch on the
    critic's stream (:func:`value_loss_fwd_bwd`) keeps its two sums in ``value_rows``: two launches on two streams must not
    read-modify-write the same words."""

    __slots__ = ("storage", "rows", "value_rows", "value_armed", "value_weight", "policy_has_value", "B", "A", "D", "weights", "blocks",
                 "armed")

    def __init__(self, B: int, A: int, D: int, device, categorical: bool):
        self.B, self.A, self.D = B, A, D
        lib = _native.lib()
        self.blocks = int(lib.cusrl_ppo_loss_blocks(B, 0 if categorical else A))
        # (both sets of rows are windows of ONE tensor: staged for the host and reset as one piece)
        policy, value = max(int(lib.cusrl_ppo_loss_num_partials(B)), 1), max(int(lib.cusrl_value_loss_blocks(B, D)), 1)
        self.storage = torch.zeros(policy * 5 + value * 2, dtype=torch.float64, device=device)
        self.rows = self.storage[: policy * 5].view(policy, 5)
        self.value_rows = self.storage[policy * 5 :].view(value, 2)
        self.weights: tuple[float, float, float] | None = None
        self.value_weight: float | None = None
        self.armed = False  # a launch of the (policy / whole) objective has been recorded against `rows`
        self.value_armed = False  # ... of the separate value term against `value_rows`
        self.policy_has_value = True

    MAX_BLOCKS = 256  # beyond this the per-row read-modify-write and the host-side sum stop being negligible

    def sums(self) -> torch.Tensor | None:
        """The five running sums as a device tensor (no host read) and a reset of the rows; None if nothing ran."""
        if not (self.armed or self.value_armed):
            return None
        total = self.rows[: self.blocks].sum(0)
        self.rows.zero_()
        if self.value_armed:  # sums 0 (squared value error) and 4 (value) came from the separate launch
            value = self.value_rows.sum(0)
            self.value_rows.zero_()
            total = torch.stack((value[0], total[1], total[2], total[3], value[1]))
        return total

    def stage(self):
        """``(storage, decode)`` for a batched host read (``Metrics._stage_pending`` snapshots and zeroes ``storage``):
        ``decode(host values of storage)`` gives what :meth:`metrics` gives, under the flags and weights in force NOW."""
        if not (self.armed or self.value_armed):
            return None
        policy_rows, blocks = self.rows.shape[0], self.blocks
        frozen = (self.armed, self.value_armed, self.policy_has_value, self.weights, self.value_weight)

        def decode(host):
            import numpy as np

            flat = np.asarray(host, dtype=np.float64)
            sums = flat[: policy_rows * 5].reshape(policy_rows, 5)[:blocks].sum(0)
            if frozen[1]:
                value = flat[policy_rows * 5 :].reshape(-1, 2).sum(0)
                sums[0], sums[4] = value[0], value[1]
            return self._metrics(sums.tolist(), *frozen)

        return self.storage, decode

    def drain(self, replays: int) -> dict[str, tuple[float, int]] | None:
        """``{metric: (sum over replays of the per-step mean, samples per step)}`` and a reset of the rows."""
        if replays <= 0 or (total := self.sums()) is None:
            return None
        return self.metrics(total.tolist())

    def metrics(self, sums: Sequence[float]) -> dict[str, tuple[float, int]]:
        return self._metrics(sums, self.armed, self.value_armed, self.policy_has_value, self.weights, self.value_weight)

    def _metrics(self, sums, armed, value_armed, policy_has_value, weights, value_weight) -> dict[str, tuple[float, int]]:
        B, D = self.B, self.D
        out: dict[str, tuple[float, int]] = {}
        if value_armed or (armed and policy_has_value):
            w_val = value_weight if value_armed else weights[0]
            out["value_loss"] = (sums[0] / (B * D) * w_val, 1)
        if armed:
            _, w_sur, w_ent = weights
            out["surrogate_loss"] = (-sums[1] / B * w_sur, 1)
            out["entropy_loss"] = (-sums[2] / B * w_ent, 1)
            out["ratio"] = (sums[3] / B, B)
            out["entropy"] = (sums[2] / B, B)
        if "value_loss" in out:
            out["value"] = (sums[4] / B, B)
        return out


def ppo_loss_fwd_bwd(
    advantage: torch.Tensor,
    old_logp: torch.Tensor,
    action: torch.Tensor,
    mean: torch.Tensor,
    std: torch.Tensor,
    ret: torch.Tensor | None,
    curr_value: torch.Tensor | None,
    old_value: torch.Tensor | None,
    *,
    clip: float,
    value_clip: float | None,
    w_sur: float,
    w_val: float,
    w_ent: float,
    want_grads: bool = True,
    deferred: DeferredLoss | None = None,
) -> dict[str, torch.Tensor]:
    """One pass: losses[0:3] = (value, surrogate, entropy) weighted losses, losses[3:6] = means of |logp ratio|, entropy
    and value (the metrics of common.py:45-49 / value.py:139-141), losses[6] = their sum, per-sample logp/entropy/ratios, and the gradients.

    ``std`` is either the ``[B, A]`` matrix or the ``[A]`` vector it repeats (a state-independent std,
    :func:`ppo_loss_accepts_std_vector`): then it is broadcast inside the kernel and ``d_std`` is the ``[A]`` gradient of
    the vector.  ``ret = curr_value = None``: the launch carries no value term (:func:`value_loss_fwd_bwd` evaluates it on
    the critic's stream); ``losses[0]``, ``losses[5]`` are 0 and there is no ``d_value``.

    ``deferred`` (a :class:`DeferredLoss` of this shape): ONE launch, no finalize — ``losses`` is absent from the result,
    the block sums accumulate in ``deferred.rows``, and with a std vector ``d_std`` comes back as
    :class:`DeferredColumns` (the blocks' column sums, reduced by ``assemble_gradients``)."""
    advantage, old_logp = _f32(advantage, "advantage"), _f32(old_logp, "action_logp")
    action, mean, std = _f32(action, "action"), _f32(mean, "mean"), _f32(std, "std")
    no_value = ret is None and curr_value is None
    if not no_value:
        ret, curr_value = _f32(ret, "return"), _f32(curr_value, "curr_value")
    A = mean.shape[-1]
    B = mean.numel() // A
    D = 0 if no_value else ret.shape[-1]
    std_vector = std.dim() == 1 and B != 1
    if std_vector and not (std.numel() == A and ppo_loss_accepts_std_vector(A)):
        raise ValueError("ppo_loss: a std vector must have one entry per action dim (and the action width a multiple of 4, <= 32)")
    if advantage.numel() != B or old_logp.numel() != B or action.shape != mean.shape or (not std_vector and std.numel() != B * A):
        raise ValueError("ppo_loss: inconsistent batch shapes")
    if no_value:
        value_clip = old_value = None
    elif ret.numel() != B * D or curr_value.shape != ret.shape:
        raise ValueError("ppo_loss: return / value shapes differ")
    if value_clip is not None:
        if old_value is None:
            raise ValueError("ppo_loss: the clipped value loss needs the old value")
        old_value = _f32(old_value, "value")
    if deferred is not None and (deferred.B, deferred.A) != (B, A) or (deferred is not None and not no_value and deferred.D != D):
        raise ValueError("ppo_loss: the deferred-loss rows belong to another minibatch shape")
    dev = mean.device
    lib = _native.lib()
    out = {
        "logp": torch.empty(advantage.shape, dtype=torch.float32, device=dev),
        "entropy": torch.empty(advantage.shape, dtype=torch.float32, device=dev),
        "logp_ratio": torch.empty(advantage.shape, dtype=torch.float32, device=dev),
        "ratio": torch.empty(advantage.shape, dtype=torch.float32, device=dev),
    }
    if deferred is None:
        out["losses"] = torch.empty(7, dtype=torch.float32, device=dev)  # 3 weighted losses, 3 metric means, total
    defer_std = deferred is not None and std_vector and want_grads
    if want_grads:
        out["d_mean"] = torch.empty_like(mean)
        if not no_value:
            out["d_value"] = torch.empty_like(curr_value)
        if not defer_std:
            out["d_std"] = torch.empty_like(std)
    if deferred is None:
        partials = torch.empty((int(lib.cusrl_ppo_loss_num_partials(B)), 5), dtype=torch.float64, device=dev)
    else:
        partials = deferred.rows
        deferred.weights, deferred.armed, deferred.policy_has_value = (float(w_val), float(w_sur), float(w_ent)), True, not no_value
    std_partials = (torch.empty((int(lib.cusrl_ppo_loss_std_partial_rows(B)), A), dtype=torch.float32, device=dev)
                    if std_vector and want_grads else None)
    if defer_std:
        out["d_std"] = DeferredColumns(std_partials, int(lib.cusrl_ppo_loss_blocks(B, A)), A, 0, A)

    def ptr(name):
        value = out.get(name)
        return value.data_ptr() if isinstance(value, torch.Tensor) else None

    _observed(
        "cusrl_ppo_loss_fwd_bwd",
        lambda: B * (8 + (8 if std_vector else 12) * A + 8 * D + ((4 if std_vector else 8) * A + 4 * D if want_grads else 0) + 16
                     + (4 * D if value_clip is not None else 0)),
        lambda: lib.cusrl_ppo_loss_fwd_bwd(
            advantage.data_ptr(), old_logp.data_ptr(), action.data_ptr(), mean.data_ptr(), std.data_ptr(),
            None if no_value else ret.data_ptr(), None if no_value else curr_value.data_ptr(),
            None if old_value is None or value_clip is None else old_value.data_ptr(),
            B, A, D, float(clip), -1.0 if value_clip is None else float(value_clip), float(w_sur), float(w_val), float(w_ent),
            ptr("losses"), ptr("logp"), ptr("entropy"), ptr("logp_ratio"), ptr("ratio"), ptr("d_mean"), ptr("d_std"), ptr("d_value"),
            partials.data_ptr(), 1 if std_vector else B, None if std_partials is None else std_partials.data_ptr(),
            LOSS_DEFER if deferred is not None else 0, _stream(),
        ),
    )
    return out


def ppo_loss_categorical_fwd_bwd(
    advantage: torch.Tensor,
    old_logp: torch.Tensor,
    action: torch.Tensor,
    logits: torch.Tensor,
    ret: torch.Tensor,
    curr_value: torch.Tensor,
    old_value: torch.Tensor | None,
    *,
    clip: float,
    value_clip: float | None,
    w_sur: float,
    w_val: float,
    w_ent: float,
    want_grads: bool = True,
    deferred: DeferredLoss | None = None,
) -> dict[str, torch.Tensor]:
    """:func:`ppo_loss_fwd_bwd` for one-hot categorical policies (``action`` one-hot ``[B, A]``, ``logits [B, A]``):
    same ``losses`` layout and per-sample outputs, gradients ``d_logits`` / ``d_value``; ``deferred`` as there."""
    advantage, old_logp = _f32(advantage, "advantage"), _f32(old_logp, "action_logp")
    action, logits = _f32(action, "action"), _f32(logits, "logits")
    no_value = ret is None and curr_value is None  # (the value term from value_loss_fwd_bwd, see ppo_loss_fwd_bwd)
    if not no_value:
        ret, curr_value = _f32(ret, "return"), _f32(curr_value, "curr_value")
    A = logits.shape[-1]
    B = logits.numel() // A
    D = 0 if no_value else ret.shape[-1]
    if advantage.numel() != B or old_logp.numel() != B or action.shape != logits.shape:
        raise ValueError("ppo_loss_categorical: inconsistent batch shapes")
    if no_value:
        value_clip = old_value = None
    elif ret.numel() != B * D or curr_value.shape != ret.shape:
        raise ValueError("ppo_loss_categorical: return / value shapes differ")
    if value_clip is not None:
        if old_value is None:
            raise ValueError("ppo_loss_categorical: the clipped value loss needs the old value")
        old_value = _f32(old_value, "value")
    if deferred is not None and ((deferred.B, deferred.A) != (B, A) or (not no_value and deferred.D != D)):
        raise ValueError("ppo_loss_categorical: the deferred-loss rows belong to another minibatch shape")
    dev = logits.device
    lib = _native.lib()
    out = {name: torch.empty(advantage.shape, dtype=torch.float32, device=dev) for name in ("logp", "entropy", "logp_ratio", "ratio")}
    if deferred is None:
        out["losses"] = torch.empty(7, dtype=torch.float32, device=dev)
        partials = torch.empty((int(lib.cusrl_ppo_loss_num_partials(B)), 5), dtype=torch.float64, device=dev)
    else:
        partials = deferred.rows
        deferred.weights, deferred.armed, deferred.policy_has_value = (float(w_val), float(w_sur), float(w_ent)), True, not no_value
    if want_grads:
        out["d_logits"] = torch.empty_like(logits)
        if not no_value:
            out["d_value"] = torch.empty_like(curr_value)

    def ptr(name):
        return out[name].data_ptr() if name in out else None

    _observed(
        "cusrl_ppo_loss_categorical_fwd_bwd",
        lambda: B * (8 + 8 * A + 8 * D + ((4 * A + 4 * D) if want_grads else 0) + 16 + (4 * D if value_clip is not None else 0)),
        lambda: lib.cusrl_ppo_loss_categorical_fwd_bwd(
            advantage.data_ptr(), old_logp.data_ptr(), action.data_ptr(), logits.data_ptr(),
            None if no_value else ret.data_ptr(), None if no_value else curr_value.data_ptr(),
            None if old_value is None or value_clip is None else old_value.data_ptr(), B, A, D, float(clip),
            -1.0 if value_clip is None else float(value_clip), float(w_sur), float(w_val), float(w_ent), ptr("losses"), ptr("logp"),
            ptr("entropy"), ptr("logp_ratio"), ptr("ratio"), ptr("d_logits"), ptr("d_value"), partials.data_ptr(),
            LOSS_DEFER if deferred is not None else 0, _stream(),
        ),
    )
    return out


def value_loss_fwd_bwd(ret: torch.Tensor, curr_value: torch.Tensor, old_value: torch.Tensor | None, *, value_clip: float | None,
                       w_val: float, want_grad: bool = True, deferred: DeferredLoss | None = None) -> dict[str, torch.Tensor]:
    """The value term alone (value.py:85-89,121-137), forward and backward in one launch on the CURRENT stream:
    ``losses`` = (weighted value loss, mean of ``curr_value.sum(-1)``) and ``d_value``.  ``deferred``: no finalize launch,
    the block sums accumulate in ``deferred.value_rows`` (and ``losses`` is absent)."""
    ret, curr_value = _f32(ret, "return"), _f32(curr_value, "curr_value")
    if curr_value.shape != ret.shape or ret.dim() < 1:
        raise ValueError("value_loss: return / value shapes differ")
    D = ret.shape[-1]
    B = ret.numel() // max(D, 1)
    if value_clip is not None:
        if old_value is None:
            raise ValueError("value_loss: the clipped value loss needs the old value")
        old_value = _f32(old_value, "value")
        if old_value.numel() != ret.numel():
            raise ValueError("value_loss: return / old value shapes differ")
    if deferred is not None and (deferred.B, deferred.D) != (B, D):
        raise ValueError("value_loss: the deferred-loss rows belong to another minibatch shape")
    dev, lib = ret.device, _native.lib()
    out: dict[str, torch.Tensor] = {}
    if want_grad:
        out["d_value"] = torch.empty_like(curr_value)
    if deferred is None:
        out["losses"] = torch.empty(2, dtype=torch.float32, device=dev)
        partials = torch.empty((max(int(lib.cusrl_value_loss_blocks(B, D)), 1), 2), dtype=torch.float64, device=dev)
    else:
        partials = deferred.value_rows
        deferred.value_armed, deferred.value_weight = True, float(w_val)
    _observed(
        "cusrl_value_loss_fwd_bwd",
        lambda: B * D * (8 + (4 if want_grad else 0) + (4 if value_clip is not None else 0)),
        lambda: lib.cusrl_value_loss_fwd_bwd(
            ret.data_ptr(), curr_value.data_ptr(), None if value_clip is None else old_value.data_ptr(), B, D,
            -1.0 if value_clip is None else float(value_clip), float(w_val),
            out["losses"].data_ptr() if deferred is None else None, out["d_value"].data_ptr() if want_grad else None,
            partials.data_ptr(), LOSS_DEFER if deferred is not None else 0, _stream()),
    )
    return out


def ppo_loss_accepts_std_vector(action_dim: int) -> bool:
    """The row-vector form of ``std`` exists for the 16-byte-chunk layout of the loss kernel."""
    return action_dim % 4 == 0 and action_dim // 4 <= 8


# ------------------------------------------------------------------------------------------------ rollout side
def normal_sample_logp(mean: torch.Tensor, std: torch.Tensor, eps: torch.Tensor, repeat_std: bool = False,
                       mean_bias: torch.Tensor | None = None):
    """``action = mean + eps * std`` and ``log_prob(action).sum(-1, keepdim=True)`` in one launch
    (cusrl/nn/module/distribution.py:198-205).  ``std`` may be the ``[A]`` vector a state-independent std repeats for every
    row; with ``repeat_std`` the launch also writes that repeated ``[B, A]`` matrix (what ``param.repeat(B, 1)`` gives,
    distribution.py:241-243) and returns it as a third value.  ``mean_bias`` ([A]): ``mean`` is the head's product without
    its bias; the launch adds it and returns the finished mean as the last value."""
    mean, std, eps = _f32(mean, "mean"), _f32(std, "std"), _f32(eps, "eps")
    A = mean.shape[-1]
    B = mean.numel() // A
    vector = std.dim() == 1 and std.numel() == A
    if (not vector and std.shape != mean.shape) or eps.shape != mean.shape:
        raise ValueError("normal_sample_logp: shape mismatch")
    action = torch.empty_like(mean)
    logp = torch.empty(mean.shape[:-1] + (1,), dtype=torch.float32, device=mean.device)
    repeated = torch.empty_like(mean) if (repeat_std and vector) else None
    finished = None
    if mean_bias is not None:
        mean_bias = _f32(mean_bias, "mean_bias")
        if mean_bias.numel() != A:
            raise ValueError("normal_sample_logp: one bias per action dim is required")
        finished = torch.empty_like(mean)
    _observed(
        "cusrl_normal_sample_logp",
        lambda: B * ((12 if vector else 16) * A + 4 + (4 * A if repeated is not None else 0) + (4 * A if finished is not None else 0)),
        lambda: _native.lib().cusrl_normal_sample_logp(mean.data_ptr(), std.data_ptr(), eps.data_ptr(), action.data_ptr(), logp.data_ptr(),
                                                       B, A, 1 if vector else B, None if repeated is None else repeated.data_ptr(),
                                                       None if finished is None else mean_bias.data_ptr(),
                                                       None if finished is None else finished.data_ptr(), _stream()),
    )
    result = (action, logp)
    if repeat_std:
        result += (repeated if repeated is not None else std,)
    if finished is not None:
        result += (finished,)
    return result


def categorical_sample_logp(logits: torch.Tensor, noise: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """One-hot sample of ``softmax(logits)`` and its log-prob in one launch (cusrl/nn/module/distribution.py:332-366):
    ``argmax(softmax(logits) / noise)`` with ``noise ~ Exp(1)`` — torch.multinomial's own single-draw rule."""
    logits, noise = _f32(logits, "logits"), _f32(noise, "noise")
    if noise.shape != logits.shape:
        raise ValueError("categorical_sample_logp: shape mismatch")
    A = logits.shape[-1]
    B = logits.numel() // A
    action = torch.empty_like(logits)
    logp = torch.empty(logits.shape[:-1] + (1,), dtype=torch.float32, device=logits.device)
    _observed(
        "cusrl_categorical_sample_logp",
        lambda: B * (12 * A + 4),
        lambda: _native.lib().cusrl_categorical_sample_logp(logits.data_ptr(), noise.data_ptr(), action.data_ptr(), logp.data_ptr(), B, A, _stream()),
    )
    return action, logp


def gru_gates_forward(gi: torch.Tensor, gh: torch.Tensor, b_hh: torch.Tensor | None, h: torch.Tensor, out: torch.Tensor,
                      lengths: torch.Tensor | None, t: int) -> None:
    """One GRU time step's gate pass (``cusrl_gru_gates_fwd``): ``h`` [B, H] is advanced in place, ``out`` [B, H] gets the
    step's output; ``gi`` / ``gh`` are the [B, 3H] projections.  Called once per step of a sequence: the caller
    (nn/gru.py) guarantees contiguous fp32 device tensors, only shapes are checked here."""
    B, H = h.shape
    if gi.shape != (B, 3 * H) or gh.shape != (B, 3 * H) or out.shape != (B, H):
        raise ValueError("gru_gates_forward: shape mismatch")
    check(_native.lib().cusrl_gru_gates_fwd(gi.data_ptr(), gh.data_ptr(), None if b_hh is None else b_hh.data_ptr(),
                                            h.data_ptr(), out.data_ptr(), None if lengths is None else lengths.data_ptr(),
                                            t, B, H, _stream()), "cusrl_gru_gates_fwd")


def gru_gates_backward(gi: torch.Tensor, gh: torch.Tensor, b_hh: torch.Tensor | None, h_prev: torch.Tensor,
                       d_out: torch.Tensor | None, dh: torch.Tensor, lengths: torch.Tensor | None, t: int,
                       bias_partials: torch.Tensor | None = None) -> None:
    """Backward of :func:`gru_gates_forward`, in place: ``gi`` / ``gh`` become their gradients, ``dh`` (the gradient that
    arrived from step t + 1) becomes the direct-path gradient of ``h_prev`` (``cusrl_gru_gates_bwd``).  With
    ``bias_partials`` (``[gru_bias_partial_rows(B), 4H]``) every block of 16 rows also leaves the column sums of the gate
    gradients it wrote — {d_r, d_z, d_n, d_q} — for the bias gradients (``cusrl_gru_gates_bwd_bias``)."""
    B, H = dh.shape
    if gi.shape != (B, 3 * H) or gh.shape != (B, 3 * H) or h_prev.shape != (B, H):
        raise ValueError("gru_gates_backward: shape mismatch")
    if bias_partials is not None:
        if bias_partials.shape != (gru_bias_partial_rows(B), 4 * H) or not bias_partials.is_contiguous():
            raise ValueError("gru_gates_backward: 'bias_partials' must be a contiguous [ceil(B / 16), 4H] tensor")
        check(_native.lib().cusrl_gru_gates_bwd_bias(gi.data_ptr(), gh.data_ptr(), None if b_hh is None else b_hh.data_ptr(),
                                                     h_prev.data_ptr(), None if d_out is None else d_out.data_ptr(), dh.data_ptr(),
                                                     None if lengths is None else lengths.data_ptr(), t, B, H,
                                                     bias_partials.data_ptr(), _stream()), "cusrl_gru_gates_bwd_bias")
        return
    check(_native.lib().cusrl_gru_gates_bwd(gi.data_ptr(), gh.data_ptr(), None if b_hh is None else b_hh.data_ptr(),
                                            h_prev.data_ptr(), None if d_out is None else d_out.data_ptr(), dh.data_ptr(),
                                            None if lengths is None else lengths.data_ptr(), t, B, H, _stream()),
          "cusrl_gru_gates_bwd")


def gru_bias_partials_supported(H: int, gi, gh, b_hh, h_prev, d_out, dh, bias_partials=None) -> bool:
    """Can :func:`gru_gates_backward` fold the bias gradients in?  Asked of the library (``cusrl_gru_bias_supported``: pointer
    alignment + column chunks that tile a 256-thread block); ``bias_partials=None``: a fresh allocation (256-byte aligned)."""
    ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    return bool(_native.lib().cusrl_gru_bias_supported(H, ptr(gi), ptr(gh), ptr(b_hh), ptr(h_prev), ptr(d_out), ptr(dh),
                                                       256 if bias_partials is None else bias_partials.data_ptr()))


def gru_bias_partial_rows(B: int) -> int:
    return int(_native.lib().cusrl_gru_bias_partial_rows(B))


def lstm_gates_forward(gi: torch.Tensor, gh: torch.Tensor, b_hh: torch.Tensor | None, h: torch.Tensor, c: torch.Tensor,
                       out: torch.Tensor, c_saved: torch.Tensor | None, lengths: torch.Tensor | None, t: int) -> None:
    """One LSTM time step's gate pass (``cusrl_lstm_gates_fwd``): ``h`` / ``c`` [B, H] advance in place, ``out`` gets the
    step's output; with ``c_saved`` (training) the new cell state is stored there and ``gi`` [B, 4H] is overwritten with
    the summed pre-activations the backward pass consumes."""
    B, H = h.shape
    if gi.shape != (B, 4 * H) or gh.shape != (B, 4 * H) or out.shape != (B, H) or c.shape != (B, H):
        raise ValueError("lstm_gates_forward: shape mismatch")
    check(_native.lib().cusrl_lstm_gates_fwd(gi.data_ptr(), gh.data_ptr(), None if b_hh is None else b_hh.data_ptr(),
                                             h.data_ptr(), c.data_ptr(), out.data_ptr(),
                                             None if c_saved is None else c_saved.data_ptr(),
                                             None if lengths is None else lengths.data_ptr(), t, B, H, _stream()),
          "cusrl_lstm_gates_fwd")


def lstm_gates_backward(pre: torch.Tensor, c_prev: torch.Tensor, c_next: torch.Tensor, d_out: torch.Tensor | None,
                        dh: torch.Tensor, dc: torch.Tensor, lengths: torch.Tensor | None, t: int) -> None:
    """Backward of :func:`lstm_gates_forward`, in place: ``pre`` becomes its gradient, ``dc`` the gradient of the previous
    cell state, ``dh`` the part of the state gradient that bypasses the step (``cusrl_lstm_gates_bwd``)."""
    B, H = dh.shape
    if pre.shape != (B, 4 * H) or c_prev.shape != (B, H) or c_next.shape != (B, H) or dc.shape != (B, H):
        raise ValueError("lstm_gates_backward: shape mismatch")
    check(_native.lib().cusrl_lstm_gates_bwd(pre.data_ptr(), c_prev.data_ptr(), c_next.data_ptr(),
                                             None if d_out is None else d_out.data_ptr(), dh.data_ptr(), dc.data_ptr(),
                                             None if lengths is None else lengths.data_ptr(), t, B, H, _stream()),
          "cusrl_lstm_gates_bwd")


def rnn_cell_forward(gi: torch.Tensor, gh: torch.Tensor, b_hh: torch.Tensor | None, h: torch.Tensor, out: torch.Tensor,
                     lengths: torch.Tensor | None, t: int, relu: bool) -> None:
    """One ``nn.RNN`` time step: ``h = act(gi + gh + b_hh)`` in place, ``out`` = the step's output (``cusrl_rnn_cell_fwd``)."""
    B, H = h.shape
    if gi.shape != (B, H) or gh.shape != (B, H) or out.shape != (B, H):
        raise ValueError("rnn_cell_forward: shape mismatch")
    check(_native.lib().cusrl_rnn_cell_fwd(gi.data_ptr(), gh.data_ptr(), None if b_hh is None else b_hh.data_ptr(),
                                           h.data_ptr(), out.data_ptr(), None if lengths is None else lengths.data_ptr(),
                                           t, B, H, int(relu), _stream()), "cusrl_rnn_cell_fwd")


def rnn_cell_backward(d_pre: torch.Tensor, out: torch.Tensor, d_out: torch.Tensor | None, dh: torch.Tensor,
                      lengths: torch.Tensor | None, t: int, relu: bool) -> None:
    """Backward of :func:`rnn_cell_forward`: ``d_pre`` receives the pre-activation gradient, ``dh`` keeps what bypasses the
    step (``cusrl_rnn_cell_bwd``)."""
    B, H = dh.shape
    if d_pre.shape != (B, H) or out.shape != (B, H):
        raise ValueError("rnn_cell_backward: shape mismatch")
    check(_native.lib().cusrl_rnn_cell_bwd(d_pre.data_ptr(), out.data_ptr(), None if d_out is None else d_out.data_ptr(),
                                           dh.data_ptr(), None if lengths is None else lengths.data_ptr(), t, B, H,
                                           int(relu), _stream()), "cusrl_rnn_cell_bwd")


def episode_stats(reward, done, episode_rew, episode_len, ring_rew, ring_len, num_episodes, step_reward_sum, parity: int) -> None:
    """One-launch ``EnvironmentStats.track_step`` + ``track_episode`` (cusrl/template/trainer.py:54-76), no host sync.
    ``num_episodes`` is the double-buffered int64[2] counter: read at ``parity``, written at ``parity ^ 1``."""
    if num_episodes.numel() != 2 or num_episodes.dtype != torch.int64:
        raise TypeError("episode_stats: 'num_episodes' must be the int64[2] double-buffered counter")
    reward = _f32(reward, "reward")
    done = _flag(done, "done")
    N, D = reward.shape
    _observed(
        "cusrl_episode_stats",
        lambda: N * (12 * D + 9),
        lambda: _native.lib().cusrl_episode_stats(
            reward.data_ptr(), done.data_ptr(), episode_rew.data_ptr(), episode_len.data_ptr(), ring_rew.data_ptr(),
            ring_len.data_ptr(), num_episodes.data_ptr(), step_reward_sum.data_ptr(), N, D, ring_len.numel(), int(parity), _stream(),
        ),
    )


def step_epilogue(reward, terminated, truncated, done_out, episode_rew, episode_len, ring_rew, ring_len, num_episodes,
                  step_reward_sum, indices_out, count_out, parity: int) -> None:
    """``done = terminated | truncated`` + episode statistics + ordered finished-env indices and their count in ONE launch
    (``cusrl_step_epilogue``); ``count_out`` may be pinned host memory (:class:`HostCounter`)."""
    reward = _f32(reward, "reward")
    terminated, truncated = _flag(terminated, "terminated"), _flag(truncated, "truncated")
    N, D = reward.shape
    if terminated.numel() != N or truncated.numel() != N or done_out.numel() != N or indices_out.numel() < N:
        raise ValueError("step_epilogue: inconsistent sizes")
    if episode_rew.shape != (N, D) or episode_len.numel() != N or ring_rew.shape[-1] != D or step_reward_sum.numel() != D:
        raise ValueError(f"step_epilogue: the reward is [{N}, {D}] but the accumulators were built for "
                         f"{tuple(episode_rew.shape)} (an env returning another channel count than its spec says?)")
    _observed(
        "cusrl_step_epilogue",
        lambda: N * (12 * D + 11),
        lambda: _native.lib().cusrl_step_epilogue(
            reward.data_ptr(), terminated.data_ptr(), truncated.data_ptr(), done_out.data_ptr(), episode_rew.data_ptr(),
            episode_len.data_ptr(), ring_rew.data_ptr(), ring_len.data_ptr(), num_episodes.data_ptr(), step_reward_sum.data_ptr(),
            indices_out.data_ptr(), count_out.data_ptr(), N, D, ring_len.numel(), int(parity), _stream(),
        ),
    )


class PendingStepEpilogue:
    """A step epilogue whose launch has been handed to the buffer push of the same env step (``cusrl_step_epilogue_push``:
    ONE launch for both).  ``launch()`` issues it on its own — the push could not take it."""

    __slots__ = ("args", "done_out")

    def __init__(self, reward, terminated, truncated, done_out, episode_rew, episode_len, ring_rew, ring_len, num_episodes,
                 step_reward_sum, indices_out, count_out, parity: int):
        self.args = (reward, terminated, truncated, done_out, episode_rew, episode_len, ring_rew, ring_len, num_episodes,
                     step_reward_sum, indices_out, count_out, parity)
        self.done_out = done_out

    def launch(self) -> None:
        step_epilogue(*self.args)

    def launch_with_push(self, table, count: int, done_field: int, cursor: int, parallelism: int) -> None:
        (reward, terminated, truncated, done_out, episode_rew, episode_len, ring_rew, ring_len, num_episodes, step_reward_sum,
         indices_out, count_out, parity) = self.args
        reward = _f32(reward, "reward")
        terminated, truncated = _flag(terminated, "terminated"), _flag(truncated, "truncated")
        N, D = reward.shape
        if terminated.numel() != N or truncated.numel() != N or done_out.numel() != N or indices_out.numel() < N or parallelism != N:
            raise ValueError("step_epilogue_push: inconsistent sizes")
        _observed(
            "cusrl_step_epilogue_push",
            lambda: N * (12 * D + 11) + sum(2 * parallelism * table[i].row_bytes for i in range(count)),
            lambda: _native.lib().cusrl_step_epilogue_push(
                reward.data_ptr(), terminated.data_ptr(), truncated.data_ptr(), done_out.data_ptr(), episode_rew.data_ptr(),
                episode_len.data_ptr(), ring_rew.data_ptr(), ring_len.data_ptr(), num_episodes.data_ptr(), step_reward_sum.data_ptr(),
                indices_out.data_ptr(), count_out.data_ptr(), N, D, ring_len.numel(), int(parity), table, count, done_field,
                cursor, _stream()),
        )


# ------------------------------------------------------------------------------------------------ differentiable policy terms
def policy_terms_fwd(mean: torch.Tensor, std: torch.Tensor, action: torch.Tensor, old_logp: torch.Tensor):
    """``(logp, entropy, logp_ratio, prob_ratio)``, each ``[..., 1]``, of a Gaussian policy in ONE launch
    (common.py:29-43, distribution.py:207-213).  ``std`` is ``[..., A]`` like ``mean`` or ONE ``[A]`` vector."""
    mean, std, action, old_logp = _f32(mean, "mean"), _f32(std, "std"), _f32(action, "action"), _f32(old_logp, "old_logp")
    A = mean.shape[-1]
    B = mean.numel() // max(A, 1)
    std_rows = 1 if std.dim() == 1 else B
    if action.numel() != B * A or old_logp.numel() != B or std.numel() != std_rows * A:
        raise ValueError("policy_terms: inconsistent shapes")
    outs = [torch.empty(mean.shape[:-1] + (1,), dtype=torch.float32, device=mean.device) for _ in range(4)]
    _observed("cusrl_policy_terms_fwd", lambda: B * (12 * A + 20),
              lambda: _native.lib().cusrl_policy_terms_fwd(mean.data_ptr(), std.data_ptr(), std_rows, action.data_ptr(),
                                                           old_logp.data_ptr(), B, A, *(o.data_ptr() for o in outs), _stream()))
    return tuple(outs)


def _optional_rows(tensor, rows: int, name: str):
    if tensor is None:
        return None
    tensor = _f32(tensor, name)
    if tensor.numel() != rows:
        raise ValueError(f"policy_terms backward: '{name}' has {tensor.numel()} elements, expected {rows}")
    return tensor


def policy_terms_bwd(mean, std, action, ratio, g_logp, g_entropy, g_logp_ratio, g_ratio):
    """``(d_mean, d_std)`` from the gradients wrt the four outputs of :func:`policy_terms_fwd` (any may be None):
    one launch (+ the one-block column-sum finalize for a std vector)."""
    mean, std, action = _f32(mean, "mean"), _f32(std, "std"), _f32(action, "action")
    A = mean.shape[-1]
    B = mean.numel() // max(A, 1)
    std_rows = 1 if std.dim() == 1 else B
    grads = [_optional_rows(g, B, n) for g, n in ((g_logp, "g_logp"), (g_entropy, "g_entropy"), (g_logp_ratio, "g_logp_ratio"),
                                                   (g_ratio, "g_ratio"))]
    ratio = _optional_rows(ratio, B, "ratio")
    d_mean, d_std = torch.empty_like(mean), torch.empty_like(std)
    lib = _native.lib()
    vector = std_rows == 1 and B != 1
    partials = (torch.empty((max(int(lib.cusrl_policy_terms_std_partial_rows(B)), 1), A), dtype=torch.float32, device=mean.device)
                if vector else None)
    ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    _observed("cusrl_policy_terms_bwd", lambda: B * (20 * A + 20),
              lambda: lib.cusrl_policy_terms_bwd(mean.data_ptr(), std.data_ptr(), std_rows, action.data_ptr(), ptr(ratio),
                                                 *(ptr(g) for g in grads), B, A, d_mean.data_ptr(), d_std.data_ptr(),
                                                 ptr(partials), _stream()))
    return d_mean, d_std


def categorical_terms_fwd(logits: torch.Tensor, action: torch.Tensor, old_logp: torch.Tensor):
    """The same four terms for a one-hot categorical policy (distribution.py:354-362)."""
    logits, action, old_logp = _f32(logits, "logits"), _f32(action, "action"), _f32(old_logp, "old_logp")
    A = logits.shape[-1]
    B = logits.numel() // max(A, 1)
    if action.numel() != B * A or old_logp.numel() != B:
        raise ValueError("categorical_terms: inconsistent shapes")
    outs = [torch.empty(logits.shape[:-1] + (1,), dtype=torch.float32, device=logits.device) for _ in range(4)]
    _observed("cusrl_categorical_terms_fwd", lambda: B * (8 * A + 20),
              lambda: _native.lib().cusrl_categorical_terms_fwd(logits.data_ptr(), action.data_ptr(), old_logp.data_ptr(), B, A,
                                                                *(o.data_ptr() for o in outs), _stream()))
    return tuple(outs)


def categorical_terms_bwd(logits, action, ratio, g_logp, g_entropy, g_logp_ratio, g_ratio):
    logits, action = _f32(logits, "logits"), _f32(action, "action")
    A = logits.shape[-1]
    B = logits.numel() // max(A, 1)
    grads = [_optional_rows(g, B, n) for g, n in ((g_logp, "g_logp"), (g_entropy, "g_entropy"), (g_logp_ratio, "g_logp_ratio"),
                                                   (g_ratio, "g_ratio"))]
    ratio = _optional_rows(ratio, B, "ratio")
    d_logits = torch.empty_like(logits)
    ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    _observed("cusrl_categorical_terms_bwd", lambda: B * (12 * A + 20),
              lambda: _native.lib().cusrl_categorical_terms_bwd(logits.data_ptr(), action.data_ptr(), ptr(ratio),
                                                                *(ptr(g) for g in grads), B, A, d_logits.data_ptr(), _stream()))
    return d_logits


# ------------------------------------------------------------------------------------------------ policy statistics
def policy_stats(old_mean: torch.Tensor, old_std: torch.Tensor, new_mean: torch.Tensor, new_std: torch.Tensor,
                 action: torch.Tensor, old_logp: torch.Tensor, advantage: torch.Tensor) -> torch.Tensor:
    """``[mean KL(old || new), mean advantage * exp(logp_new(action) - old_logp), mean new_std]`` of a Gaussian policy
    over a batch (cusrl/hook/on_policy/stats.py:28-40) — a 3-element device tensor from one pass."""
    tensors = [_f32(t, n) for t, n in ((old_mean, "old_mean"), (old_std, "old_std"), (new_mean, "new_mean"), (new_std, "new_std"),
                                       (action, "action"), (old_logp, "old_logp"), (advantage, "advantage"))]
    A = new_mean.shape[-1]
    B = new_mean.numel() // A
    D = advantage.numel() // B
    if any(t.numel() != B * A for t in tensors[:5]) or tensors[5].numel() != B or tensors[6].numel() != B * D:
        raise ValueError("policy_stats: inconsistent shapes")
    lib = _native.lib()
    dev = new_mean.device
    partials = torch.empty((max(int(lib.cusrl_policy_stats_num_partials(B)), 1), 3), dtype=torch.float64, device=dev)
    out = torch.empty(3, dtype=torch.float32, device=dev)
    check(lib.cusrl_policy_stats(*(t.data_ptr() for t in tensors), B, A, D, partials.data_ptr(), out.data_ptr(), _stream()),
          "cusrl_policy_stats")
    return out


def categorical_policy_stats(old_logits: torch.Tensor, new_logits: torch.Tensor, action: torch.Tensor, old_logp: torch.Tensor,
                             advantage: torch.Tensor) -> torch.Tensor:
    """``[mean KL(old || new), mean advantage * exp(logp_new(action) - old_logp), 0]`` of a one-hot categorical policy over
    a batch — the discrete-action form of :func:`policy_stats`."""
    tensors = [_f32(t, n) for t, n in ((old_logits, "old_logits"), (new_logits, "new_logits"), (action, "action"),
                                       (old_logp, "old_logp"), (advantage, "advantage"))]
    A = new_logits.shape[-1]
    B = new_logits.numel() // A
    D = advantage.numel() // B
    if any(t.numel() != B * A for t in tensors[:3]) or tensors[3].numel() != B or tensors[4].numel() != B * D:
        raise ValueError("categorical_policy_stats: inconsistent shapes")
    lib = _native.lib()
    dev = new_logits.device
    partials = torch.empty((max(int(lib.cusrl_policy_stats_num_partials(B)), 1), 3), dtype=torch.float64, device=dev)
    out = torch.empty(3, dtype=torch.float32, device=dev)
    check(lib.cusrl_categorical_policy_stats(*(t.data_ptr() for t in tensors), B, A, D, partials.data_ptr(), out.data_ptr(), _stream()),
          "cusrl_categorical_policy_stats")
    return out


# ------------------------------------------------------------------------------------------------ captured-graph census
def graph_census(graph: "torch.cuda.CUDAGraph") -> dict:
    """What a captured hipGraph is made of (``cusrl_graph_census``): ``{"kernel": n, "memcpy": n, "memset": n, "other": n,
    "names": [mangled kernel names in node order]}``.  ``graph`` must have been created with ``keep_graph=True``."""
    import ctypes

    lib = _native.lib()
    raw = ctypes.c_void_p(int(graph.raw_cuda_graph()))
    counts = (ctypes.c_int64 * 16)()
    need = ctypes.c_int64(0)
    capacity = 1 << 16
    while True:
        names = ctypes.create_string_buffer(capacity)
        check(lib.cusrl_graph_census(raw, counts, 16, names, capacity, ctypes.byref(need)), "cusrl_graph_census")
        if need.value <= capacity:
            break
        capacity = need.value
    listed = names.raw[: need.value].decode(errors="replace").split("\n")[:-1]
    return {"kernel": counts[0], "memcpy": counts[1], "memset": counts[2], "other": sum(counts[3:]), "names": listed}


def graph_replace_memsets(graph: "torch.cuda.CUDAGraph") -> int:
    """Turn every memset node of a kept, not yet instantiated hipGraph into a fill-kernel node (``cusrl_graph_replace_memsets``);
    returns how many were replaced."""
    import ctypes

    replaced = ctypes.c_int64(0)
    check(_native.lib().cusrl_graph_replace_memsets(ctypes.c_void_p(int(graph.raw_cuda_graph())), ctypes.byref(replaced)),
          "cusrl_graph_replace_memsets")
    return int(replaced.value)


# ------------------------------------------------------------------------------------------------ MLP backward epilogue
def relu_backward_bias(grad_output: torch.Tensor, output: torch.Tensor | None, defer: bool = False):
    """``(grad_output * (output > 0), masked.sum(0))`` in one pass; with ``output=None`` just the column sums
    (bias gradient of a linear layer, with or without the ReLU that follows it).  ``defer``: return the column sums as
    :class:`DeferredColumns` (no finalize launch) when the layout allows; the flat gradient assembly reduces them."""
    grad_output = _f32(grad_output, "grad_output")
    H = grad_output.shape[-1]
    rows = grad_output.numel() // H
    lib = _native.lib()
    num_partials = max(int(lib.cusrl_colsum_num_partials(rows, H)), 1)
    partials = torch.empty((num_partials, H), dtype=torch.float32, device=grad_output.device)
    chunkable = H % 4 == 0 and H // 4 <= 256 and 256 % (H // 4) == 0
    defer = defer and chunkable and grad_output.data_ptr() % 16 == 0 and (output is None or output.data_ptr() % 16 == 0)
    colsum = None if defer else torch.empty(H, dtype=torch.float32, device=grad_output.device)
    if output is None:
        grad_in, out_ptr, in_ptr = grad_output, None, None
    else:
        output = _f32(output, "output")
        grad_in = torch.empty_like(grad_output)
        out_ptr, in_ptr = output.data_ptr(), grad_in.data_ptr()
    check(
        lib.cusrl_relu_bwd_colsum(grad_output.data_ptr(), out_ptr, in_ptr, partials.data_ptr(),
                                  None if colsum is None else colsum.data_ptr(), rows, H, _stream()),
        "cusrl_relu_bwd_colsum",
    )
    return grad_in, (DeferredColumns(partials, num_partials, H, 0, H) if colsum is None else colsum)


# ------------------------------------------------------------------------------------------------ first layer of an MLP
def input_layer_supported(grad_output: torch.Tensor, output: torch.Tensor | None, input: torch.Tensor, weight: torch.Tensor) -> bool:
    """Shapes and layouts ``cusrl_input_layer_bwd`` takes: fp32, contiguous, 16-byte aligned, K % 4 == 0 (<= 60), H % 64 == 0."""
    H, K = weight.shape
    tensors = [grad_output, input] + ([] if output is None else [output])
    return (bool(_native.lib().cusrl_input_layer_supported(K, H)) and input.dim() == 2 and grad_output.shape == (input.shape[0], H)
            and all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 16 == 0 for t in tensors)
            and (output is None or output.shape == grad_output.shape) and input.shape[0] > 0)


def input_layer_backward(grad_output: torch.Tensor, output: torch.Tensor | None, input: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """``(dW [H, K], db [H])`` of ``y = relu(x W^T + b)`` for an input that needs no gradient, from ONE pass over ``grad_output``,
    ``output`` (the ReLU's output; None: no activation) and ``input`` (``cusrl_input_layer_bwd``); both are windows of one
    ``[H * K + H]`` row."""
    grad_output, input = _f32(grad_output, "grad_output"), _f32(input, "input")
    rows, K = input.shape
    H = grad_output.shape[-1]
    lib = _native.lib()
    dev = input.device
    width = H * K + H
    partials = torch.empty((int(lib.cusrl_input_layer_row_blocks(rows, H)), width), dtype=torch.float32, device=dev)
    grads = torch.empty(width, dtype=torch.float32, device=dev)
    _observed(
        "cusrl_input_layer_bwd",
        lambda: rows * 4 * ((2 if output is not None else 1) * H + K) + (partials.numel() * 2 + width) * 4,
        lambda: lib.cusrl_input_layer_bwd(grad_output.data_ptr(), None if output is None else _f32(output, "output").data_ptr(),
                                          input.data_ptr(), rows, K, H, partials.data_ptr(), grads.data_ptr(), _stream()),
    )
    return grads[: H * K].view(H, K), grads[H * K :]


# ------------------------------------------------------------------------------------------------ narrow heads
_HEAD_PAD = 16


def narrow_linear_supported(in_features: int, out_features: int) -> bool:
    return bool(_native.lib().cusrl_narrow_linear_supported(in_features, out_features))


def narrow_linear_forward_supported(input: torch.Tensor, weight: torch.Tensor) -> bool:
    """A one-output head over a contiguous, 16-byte aligned fp32 ``[B, K]`` device matrix with K a power of two in 32..1024."""
    return (weight.dim() == 2 and weight.shape[0] == 1 and input.dim() == 2 and input.is_cuda and input.dtype == torch.float32
            and weight.dtype == torch.float32 and input.is_contiguous() and weight.is_contiguous() and input.shape[0] > 0
            and input.data_ptr() % 16 == 0 and weight.data_ptr() % 16 == 0 and narrow_linear_supported(weight.shape[1], 1))


def narrow_linear_forward(input: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None) -> torch.Tensor:
    """``input @ weight.T + bias`` for a ONE-output linear layer (value head, discriminator logit) in one launch
    (``cusrl_narrow_linear_fwd``: a row dot product) instead of torch's broadcast-bias copy + skinny GEMM."""
    input, weight = _f32(input, "input"), _f32(weight, "weight")
    rows, K = input.shape
    out = torch.empty((rows, 1), dtype=torch.float32, device=input.device)
    check(
        _native.lib().cusrl_narrow_linear_fwd(input.data_ptr(), weight.data_ptr(), None if bias is None else _f32(bias, "bias").data_ptr(),
                                              out.data_ptr(), rows, K, 1, _stream()),
        "cusrl_narrow_linear_fwd",
    )
    return out


def mlp2_forward_supported(input: torch.Tensor, layers) -> bool:
    """``layers`` = ``(w1, b1, w2, b2, w3, b3)`` (``b3`` may be None) of a Linear / ReLU / Linear / ReLU / Linear stack whose shapes
    the one-launch inference pass takes (``cusrl_mlp2_forward_supported``), over a contiguous fp32 ``[B, K]`` device matrix."""
    w1, b1, w2, b2, w3, b3 = layers
    tensors = [input, w1, b1, w2, b2, w3] + ([b3] if b3 is not None else [])
    if input.dim() != 2 or input.shape[0] == 0 or any(
            (not t.is_cuda) or t.dtype != torch.float32 or (not t.is_contiguous()) or t.data_ptr() % 16 for t in tensors):
        return False
    if w1.dim() != 2 or w2.dim() != 2 or w3.dim() != 2 or w1.shape[1] != input.shape[1] or w2.shape[1] != w1.shape[0] \
            or w3.shape[1] != w2.shape[0] or b1.numel() != w1.shape[0] or b2.numel() != w2.shape[0] \
            or (b3 is not None and b3.numel() != w3.shape[0]):
        return False
    return bool(_native.lib().cusrl_mlp2_forward_supported(w1.shape[1], w1.shape[0], w2.shape[0], w3.shape[0]))


def mlp2_forward(input: torch.Tensor, layers, std: torch.Tensor | None = None, eps: torch.Tensor | None = None,
                 repeat_std: bool = True):
    """``w3 relu(w2 relu(w1 x + b1) + b2) + b3`` in ONE launch (``cusrl_mlp2_forward``), no autograd: the head's output
    ``[B, out]``; with ``std`` (``[out]`` vector) and ``eps`` (``[B, out]``) the acting path's ``(action, logp [B, 1], mean,
    repeated std or None)`` — ``action = mean + eps * std`` and its log-prob as ``normal_sample_logp`` evaluates them."""
    w1, b1, w2, b2, w3, b3 = layers
    input = _f32(input, "input")
    rows, K = input.shape
    out_features = w3.shape[0]
    out = torch.empty((rows, out_features), dtype=torch.float32, device=input.device)
    sampling = eps is not None
    action = logp = repeated = None
    if sampling:
        std, eps = _f32(std, "std"), _f32(eps, "eps")
        if std.numel() != out_features or eps.shape != out.shape:
            raise ValueError("mlp2_forward: one std per output and one eps per output element are required")
        action = torch.empty_like(out)
        logp = torch.empty((rows, 1), dtype=torch.float32, device=input.device)
        repeated = torch.empty_like(out) if repeat_std else None
    ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    _observed(
        "cusrl_mlp2_forward",
        lambda: rows * 4 * (K + out_features * (4 if sampling else 1)) + 4 * (w1.numel() + w2.numel() + w3.numel()),
        lambda: _native.lib().cusrl_mlp2_forward(input.data_ptr(), rows, K, w1.data_ptr(), b1.data_ptr(), w1.shape[0], w2.data_ptr(),
                                                 b2.data_ptr(), w2.shape[0], w3.data_ptr(), ptr(b3), out_features, out.data_ptr(),
                                                 ptr(std) if sampling else None, ptr(eps), ptr(action), ptr(logp), ptr(repeated),
                                                 _stream()),
    )
    if sampling:
        return action, logp, out, repeated
    return out


def narrow_linear_backward(grad_output: torch.Tensor, input: torch.Tensor, weight: torch.Tensor,
                           need_input_grad: bool = True, relu_input: bool = False, defer: bool = False):
    """``(grad_output @ weight, grad_output.T @ input, grad_output.sum(0))`` of a linear layer with at most 16
    outputs (policy-mean / value head) in one pass over the minibatch.  With ``relu_input`` (the layer's input is a
    ReLU output) the returned grad_input is already masked by ``input > 0`` and a fourth value, its column sums
    (the bias gradient of the layer in front of the ReLU), is returned; otherwise the fourth value is None.  ``defer``:
    dW, db and the column sums come back as :class:`DeferredColumns` (no finalize launch)."""
    grad_output, input, weight = _f32(grad_output, "grad_output"), _f32(input, "input"), _f32(weight, "weight")
    O, K = weight.shape
    rows = input.shape[0]
    lib = _native.lib()
    dev = input.device
    grad_input = torch.empty_like(input) if need_input_grad else None
    width = (O + 1) * K + _HEAD_PAD
    num_partials = int(lib.cusrl_narrow_linear_num_partials(rows))
    partials = torch.empty((num_partials, width), dtype=torch.float32, device=dev)
    packed = None if defer else torch.empty(width, dtype=torch.float32, device=dev)
    check(
        lib.cusrl_narrow_linear_bwd(grad_output.data_ptr(), input.data_ptr(), weight.data_ptr(),
                                    None if grad_input is None else grad_input.data_ptr(), partials.data_ptr(),
                                    None if packed is None else packed.data_ptr(), rows, K, O, int(relu_input), _stream()),
        "cusrl_narrow_linear_bwd",
    )
    if defer:  # the three gradients stay as windows of the partial rows; cusrl_assemble_gradients sums them
        window = lambda column, numel: DeferredColumns(partials, num_partials, width, column, numel)  # noqa: E731
        return grad_input, window(0, O * K), window((O + 1) * K, O), (window(O * K, K) if relu_input else None)
    colsum = packed[O * K : (O + 1) * K] if relu_input else None
    return grad_input, packed[: O * K].view(O, K), packed[(O + 1) * K : (O + 1) * K + O], colsum


# ------------------------------------------------------------------------------------------------ gradient clipping
def clip_grad_norm_(flat_grad: torch.Tensor, max_norm: float | None) -> torch.Tensor:
    """``torch.nn.utils.clip_grad_norm_`` on one flat fp32 gradient buffer, in place: returns the pre-clip L2 norm
    (0-d device tensor); ``max_norm=None`` only measures (gradient_clipping.py:67-83)."""
    flat_grad = _f32(flat_grad, "flat_grad")
    lib = _native.lib()
    n = flat_grad.numel()
    partials = torch.empty(max(int(lib.cusrl_clip_grad_norm_num_partials(n)), 1), dtype=torch.float64, device=flat_grad.device)
    norm = torch.empty(1, dtype=torch.float32, device=flat_grad.device)
    check(
        lib.cusrl_clip_grad_norm(flat_grad.data_ptr(), n, -1.0 if max_norm is None else float(max_norm),
                                 partials.data_ptr(), norm.data_ptr(), _stream()),
        "cusrl_clip_grad_norm",
    )
    if max_norm is not None:
        _modified_in_place(flat_grad)  # scaled in place: stale squared-norm partials (FlatGradients.take_sumsq) must not survive
    return norm[0]


class DeferredColumns:
    """Column sums that have NOT been taken yet: ``splits`` partial rows of width ``row_stride`` floats, of which the
    window ``[column, column + numel)`` sums to a gradient.  The column-sum kernels (ReLU-backward + bias gradient,
    narrow-head backward) hand these out instead of running their finalize launch when the flat gradient assembly is
    going to reduce them anyway (``assemble_gradients``)."""

    __slots__ = ("partials", "splits", "row_stride", "column", "numel")

    def __init__(self, partials: torch.Tensor, splits: int, row_stride: int, column: int, numel: int):
        self.partials, self.splits, self.row_stride, self.column, self.numel = partials, splits, row_stride, column, numel

    def materialize(self) -> torch.Tensor:
        out = torch.empty(self.numel, dtype=torch.float32, device=self.partials.device)
        assemble_gradients([(self, 0, self.numel, self.splits)], out)
        return out


def sum_slabs(slabs: torch.Tensor) -> torch.Tensor:
    """``slabs.sum(0)`` of a contiguous ``[S, ...]`` fp32 stack in fixed order through ``cusrl_assemble_gradients`` (one piece):
    no ATen reduction — a global ``reduce_kernel`` brings a semaphore memset node into a captured step."""
    out = torch.empty(slabs.shape[1:], dtype=torch.float32, device=slabs.device)
    if out.numel():
        assemble_gradients([(slabs, 0, out.numel(), slabs.shape[0])], out.view(-1))  # (validates device / dtype / layout)
    return out


def assemble_gradients(pieces: Sequence[tuple], flat: torch.Tensor, want_sumsq: bool = False):
    """Fill the flat gradient buffer in one launch.  ``pieces`` = ``(src, offset, numel, splits)`` per parameter:
    ``src [splits, numel]`` slabs are summed into ``flat[offset : offset + numel]``; ``splits = 1`` copies a plain
    gradient, ``src = None`` / ``splits = 0`` writes zeros; a :class:`DeferredColumns` ``src`` is reduced over its
    partial rows (``splits`` is taken from it).  ``want_sumsq``: also return the blocks' partial sums of squares of what
    they wrote (fp64) — the squared gradient norm :func:`adam_step` turns into the clipping coefficient."""
    flat = _f32(flat, "flat")
    table = (_native.GradPiece * max(len(pieces), 1))()
    keep = []
    for slot, (src, offset, numel, splits) in zip(table, pieces):
        slot.row_stride = 0
        if isinstance(src, DeferredColumns):
            if src.numel != numel or offset < 0 or offset + numel > flat.numel():
                raise ValueError("deferred column sums do not match the parameter's slot")
            keep.append(src.partials)
            slot.src, slot.splits, slot.row_stride = src.partials.data_ptr() + 4 * src.column, src.splits, src.row_stride
        elif src is None or splits == 0:
            slot.src, slot.splits = None, 0
        else:
            src = _f32(src, "gradient piece")
            if src.numel() != splits * numel:
                raise ValueError(f"gradient piece has {src.numel()} elements, expected {splits} x {numel}")
            if offset < 0 or offset + numel > flat.numel():
                raise ValueError("gradient piece does not fit the flat buffer")
            keep.append(src)
            slot.src, slot.splits = src.data_ptr(), splits
        slot.offset, slot.numel = offset, numel
    lib = _native.lib()
    sumsq = None
    if want_sumsq:
        blocks = int(lib.cusrl_assemble_gradients_blocks(table, len(pieces)))
        if 0 < blocks <= 1 << 16:
            sumsq = torch.empty(blocks, dtype=torch.float64, device=flat.device)
    check(lib.cusrl_assemble_gradients(table, len(pieces), flat.data_ptr(), None if sumsq is None else sumsq.data_ptr(), _stream()),
          "cusrl_assemble_gradients")
    return sumsq


def grad_sumsq(flat_grad: torch.Tensor) -> torch.Tensor:
    """Block partials (double) of ``sum(grad ** 2)`` — the pending norm that :func:`adam_step` turns into the clipping
    coefficient while it streams the gradient."""
    flat_grad = _f32(flat_grad, "flat_grad")
    lib = _native.lib()
    n = flat_grad.numel()
    partials = torch.empty(max(int(lib.cusrl_clip_grad_norm_num_partials(n)), 1), dtype=torch.float64, device=flat_grad.device)
    check(lib.cusrl_grad_sumsq(flat_grad.data_ptr(), n, partials.data_ptr(), _stream()), "cusrl_grad_sumsq")
    return partials


def adam_step(param: torch.Tensor, grad: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor,
              step: torch.Tensor, lr: torch.Tensor, ticket: torch.Tensor, *, betas: tuple[float, float], eps: float,
              weight_decay: float, decoupled: bool, maximize: bool = False, clip_partials: torch.Tensor | None = None,
              max_norm: float | None = None, norm_out: torch.Tensor | None = None, norm_accumulator: torch.Tensor | None = None):
    """One Adam / AdamW step over flat fp32 buffers, in place (``step`` and ``lr`` are 1-element device tensors).
    ``norm_accumulator`` (a 1-element fp32 view): the pre-clip gradient norm is also added to it."""
    if norm_accumulator is not None:
        _f32(norm_accumulator, "norm_accumulator")
    for tensor, name in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq"), (step, "step"), (lr, "lr")):
        _f32(tensor, name)
    require_device(ticket, "ticket")
    if ticket.dtype != torch.int32 or ticket.numel() != 1:
        raise TypeError("'ticket' must be a 1-element int32 device tensor")
    n = param.numel()
    if not (grad.numel() == exp_avg.numel() == exp_avg_sq.numel() == n):
        raise ValueError("flat optimizer buffers must have the same length")
    check(
        _native.lib().cusrl_adam_step(
            param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), step.data_ptr(), lr.data_ptr(), n,
            float(betas[0]), float(betas[1]), float(eps), float(weight_decay), int(decoupled), int(maximize),
            None if clip_partials is None else clip_partials.data_ptr(), 0 if clip_partials is None else clip_partials.numel(),
            -1.0 if max_norm is None else float(max_norm), None if norm_out is None else norm_out.data_ptr(),
            None if norm_accumulator is None else norm_accumulator.data_ptr(), ticket.data_ptr(), _stream()),
        "cusrl_adam_step",
    )


def adam_step_window(param: torch.Tensor, grad: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor,
                     step: torch.Tensor, lr: torch.Tensor, ticket: torch.Tensor, *, betas: tuple[float, float], eps: float,
                     weight_decay: float, decoupled: bool, maximize: bool = False,
                     clip_partials: tuple[torch.Tensor | None, torch.Tensor | None] = (None, None), max_norm: float | None = None,
                     norm_out: torch.Tensor | None = None, norm_accumulator: torch.Tensor | None = None,
                     step_mirror: torch.Tensor | None = None):
    """:func:`adam_step` over one window of the flat buffers (``cusrl_adam_step_window``): ``clip_partials`` = the squared-norm
    partial rows of up to two gradient assemblies, summed as one array; ``step_mirror``: a second counter set to the new count."""
    for tensor, name in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq"), (step, "step"), (lr, "lr")):
        _f32(tensor, name)
    require_device(ticket, "ticket")
    if ticket.dtype != torch.int32 or ticket.numel() != 1:
        raise TypeError("'ticket' must be a 1-element int32 device tensor")
    n = param.numel()
    if not (grad.numel() == exp_avg.numel() == exp_avg_sq.numel() == n) or not all(t.is_contiguous() for t in (param, grad, exp_avg, exp_avg_sq)):
        raise ValueError("a window of the flat optimizer buffers: four contiguous stretches of the same length")
    first, second = clip_partials
    if any(t is not None and (t.dtype != torch.float64 or not t.is_cuda) for t in (first, second)):
        raise TypeError("'clip_partials' are fp64 device tensors")
    ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    check(
        _native.lib().cusrl_adam_step_window(
            param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), step.data_ptr(), lr.data_ptr(), n,
            float(betas[0]), float(betas[1]), float(eps), float(weight_decay), int(decoupled), int(maximize),
            ptr(first), 0 if first is None else first.numel(), ptr(second), 0 if second is None else second.numel(),
            -1.0 if max_norm is None else float(max_norm), ptr(norm_out), ptr(norm_accumulator), ptr(step_mirror), ticket.data_ptr(),
            _stream()),
        "cusrl_adam_step_window",
    )


def adam_norm_workspace(device) -> torch.Tensor:
    """A workspace of :func:`adam_step_normed` (0xFF bytes; one per launch that may run beside another one)."""
    return torch.full((int(_native.lib().cusrl_adam_step_normed_workspace_bytes()),), 0xFF, dtype=torch.uint8, device=device)


def adam_step_normed(param: torch.Tensor, grad: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor,
                     step: torch.Tensor, lr: torch.Tensor, ticket: torch.Tensor, *, norm_grad: torch.Tensor, workspace: torch.Tensor,
                     betas: tuple[float, float], eps: float, weight_decay: float, decoupled: bool, maximize: bool = False,
                     max_norm: float | None = None, norm_out: torch.Tensor | None = None,
                     norm_accumulator: torch.Tensor | None = None, step_mirror: torch.Tensor | None = None):
    """:func:`adam_step_window` whose launch measures ``||norm_grad||`` itself (``cusrl_adam_step_normed``): the clipping
    coefficient of a step whose gradients were averaged over the ranks after their assembly — no squared-norm launch in between."""
    for tensor, name in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq"), (step, "step"), (lr, "lr"),
                         (norm_grad, "norm_grad")):
        _f32(tensor, name)
    require_device(ticket, "ticket")
    require_device(workspace, "workspace")
    if ticket.dtype != torch.int32 or ticket.numel() != 1:
        raise TypeError("'ticket' must be a 1-element int32 device tensor")
    lib = _native.lib()
    if workspace.dtype != torch.uint8 or workspace.numel() < int(lib.cusrl_adam_step_normed_workspace_bytes()):
        raise TypeError("'workspace' comes from ops.adam_norm_workspace")
    n = param.numel()
    if not (grad.numel() == exp_avg.numel() == exp_avg_sq.numel() == n) or not all(
            t.is_contiguous() for t in (param, grad, exp_avg, exp_avg_sq, norm_grad)):
        raise ValueError("a window of the flat optimizer buffers: four contiguous stretches of the same length")
    ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    check(
        lib.cusrl_adam_step_normed(
            param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), step.data_ptr(), lr.data_ptr(), n,
            float(betas[0]), float(betas[1]), float(eps), float(weight_decay), int(decoupled), int(maximize),
            norm_grad.data_ptr(), norm_grad.numel(), workspace.data_ptr(), -1.0 if max_norm is None else float(max_norm),
            ptr(norm_out), ptr(norm_accumulator), ptr(step_mirror), ticket.data_ptr(), _stream()),
        "cusrl_adam_step_normed",
    )


# ------------------------------------------------------------------------------------------------ running statistics
def masked_col_stats(x: torch.Tensor, mask: torch.Tensor | None = None) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """``mean_var_count`` (population variance) of the rows of ``x [rows, C]`` whose ``mask`` byte is set
    (cusrl/nn/utils/normalization.py:15-50 after the ``observation[indices]`` select of observation.py:206-208);
    the count stays on the device (double[1]) — no host synchronisation."""
    x = _f32(x, "input")
    C = x.shape[-1]
    rows = x.numel() // C
    if mask is not None:
        mask = _flag(mask, "mask")
        if mask.numel() != rows:
            raise ValueError("masked_col_stats: mask must have one entry per row")
    lib = _native.lib()
    dev = x.device
    partials = torch.empty((max(int(lib.cusrl_masked_stats_num_partials(rows, C)), 1), C + 1, 2), dtype=torch.float64, device=dev)
    mean, var = torch.empty(C, dtype=torch.float32, device=dev), torch.empty(C, dtype=torch.float32, device=dev)
    count = torch.empty(1, dtype=torch.float64, device=dev)
    check(
        lib.cusrl_masked_col_stats(x.data_ptr(), None if mask is None else mask.data_ptr(), rows, C, partials.data_ptr(),
                                   mean.data_ptr(), var.data_ptr(), count.data_ptr(), _stream()),
        "cusrl_masked_col_stats",
    )
    return mean, var, count


def rms_merge_(mean, var, std, count, batch_mean, batch_var, batch_count, eps: float, max_count: float | None) -> None:
    """In-place Chan merge of batch statistics into running statistics (normalization.py:80-93, rms.py:163-167)."""
    check(
        _native.lib().cusrl_rms_merge(
            _f32(mean, "mean").data_ptr(), _f32(var, "var").data_ptr(), _f32(std, "std").data_ptr(), count.data_ptr(),
            _f32(batch_mean, "batch_mean").data_ptr(), _f32(batch_var, "batch_var").data_ptr(), batch_count.data_ptr(),
            float(eps), -1.0 if max_count is None else float(max_count), mean.numel(), _stream(),
        ),
        "cusrl_rms_merge",
    )


def rms_normalize(x: torch.Tensor, mean: torch.Tensor, std: torch.Tensor, clamp: float | None) -> torch.Tensor:
    """``((x - mean) / std).clamp(-clamp, clamp)`` as one launch (rms.py:198-203)."""
    x = _f32(x, "input")
    C = x.shape[-1]
    out = torch.empty_like(x)
    check(
        _native.lib().cusrl_rms_normalize(x.data_ptr(), _f32(mean, "mean").data_ptr(), _f32(std, "std").data_ptr(),
                                          -1.0 if clamp is None else float(clamp), out.data_ptr(), x.numel() // C, C, _stream()),
        "cusrl_rms_normalize",
    )
    return out


# ------------------------------------------------------------------------------------------------ intrinsic rewards
def rnd_reward_(reward: torch.Tensor, target: torch.Tensor, prediction: torch.Tensor, scale: float) -> torch.Tensor:
    """``reward += scale * (target - prediction).square().mean(-1, keepdim=True)`` in place, one launch; returns the
    added bonus (cusrl/hook/auxiliary/rnd.py:71-74)."""
    target, prediction = _f32(target, "target"), _f32(prediction, "prediction")
    require_device(reward, "reward")
    if reward.dtype != torch.float32 or not reward.is_contiguous() or reward.shape[-1] != 1:
        raise TypeError("rnd_reward_: reward must be a contiguous float32 [..., 1] tensor")
    K = target.shape[-1]
    rows = target.numel() // K
    if reward.numel() != rows or prediction.shape != target.shape:
        raise ValueError("rnd_reward_: shape mismatch")
    bonus = torch.empty_like(reward)
    check(_native.lib().cusrl_rnd_reward(target.data_ptr(), prediction.data_ptr(), reward.data_ptr(), bonus.data_ptr(),
                                         float(scale), rows, K, _stream()), "cusrl_rnd_reward")
    _modified_in_place(reward)
    return bonus


def amp_style_reward_(reward: torch.Tensor, logit: torch.Tensor, scale: float) -> torch.Tensor:
    """``reward += scale * -log(clamp(1 - sigmoid(logit), 1e-4))`` in place, one launch; returns the bonus
    (cusrl/hook/auxiliary/amp.py:134-136)."""
    logit = _f32(logit, "logit")
    require_device(reward, "reward")
    if reward.dtype != torch.float32 or not reward.is_contiguous() or reward.numel() != logit.numel():
        raise TypeError("amp_style_reward_: reward must be a contiguous float32 tensor matching the logits")
    bonus = torch.empty_like(reward)
    check(_native.lib().cusrl_amp_style_reward(logit.data_ptr(), reward.data_ptr(), bonus.data_ptr(), float(scale),
                                               logit.numel(), _stream()), "cusrl_amp_style_reward")
    _modified_in_place(reward)
    return bonus


def amp_style_reward_mean_(reward: torch.Tensor, logit: torch.Tensor, scale: float) -> tuple[torch.Tensor, torch.Tensor]:
    """:func:`amp_style_reward_` + the mean of the bonus (what ``agent.record(amp_reward=...)`` reduces) from ONE launch;
    returns ``(bonus, mean[1])``.  Falls back to the two-launch form beyond the single-workgroup size."""
    logit = _f32(logit, "logit")
    require_device(reward, "reward")
    if reward.dtype != torch.float32 or not reward.is_contiguous() or reward.numel() != logit.numel():
        raise TypeError("amp_style_reward_mean_: reward must be a contiguous float32 tensor matching the logits")
    if logit.numel() > (1 << 20):
        bonus = amp_style_reward_(reward, logit, scale)
        return bonus, bonus.mean().reshape(1)
    bonus = torch.empty_like(reward)
    mean = torch.empty(1, dtype=torch.float32, device=reward.device)
    check(_native.lib().cusrl_amp_style_reward_mean(logit.data_ptr(), reward.data_ptr(), bonus.data_ptr(), float(scale),
                                                    logit.numel(), mean.data_ptr(), _stream()), "cusrl_amp_style_reward_mean")
    _modified_in_place(reward)
    return bonus, mean


def amp_prepare_supported(rows: int, channels: int) -> bool:
    return 0 < channels <= 128 and 0 < rows * channels <= int(_native.lib().cusrl_amp_prepare_max_elements())


def amp_prepare(rms, *, state=None, next_state=None, columns=None, width: int | None = None, agent_raw=None, dataset=None,
                indices=None, expert_raw=None) -> tuple[torch.Tensor, torch.Tensor]:
    """AMP's ``post_step`` up to the discriminator (amp.py:112-128) as ONE C-ABI call (two launches): assemble ``state[cols] || next_state[cols]``
    (or take ``agent_raw``), fetch ``dataset[indices]`` (or take ``expert_raw``), update ``rms`` (a RunningMeanStd) with the
    agent rows, then the expert rows, normalise both.  ``columns``: int32 device vector of the selected state columns, or
    None with ``width`` = K for the first K columns.  Returns ``(agent_transition, expert_transition)``, ``[N, C]`` each."""
    if agent_raw is not None:
        agent_raw = _f32(agent_raw, "agent_raw")
        N, C = agent_raw.shape
        K = C // 2
    else:
        state, next_state = _f32(state, "state"), _f32(next_state, "next_state")
        if state.dim() != 2 or state.shape != next_state.shape:
            raise ValueError("amp_prepare: state / next_state must be [N, S] tensors of one shape")
        N = state.shape[0]
        K = int(columns.numel()) if columns is not None else int(width)
        C = 2 * K
        if columns is not None and (columns.dtype != torch.int32 or not columns.is_cuda):
            raise TypeError("amp_prepare: 'columns' must be an int32 device vector")
    if expert_raw is not None:
        expert_raw = _f32(expert_raw, "expert_raw")
        if tuple(expert_raw.shape) != (N, C):
            raise ValueError("amp_prepare: expert rows do not match the agent rows")
    else:
        dataset = _f32(dataset, "dataset")
        if indices.dtype != torch.int64 or indices.numel() != N or dataset.shape[-1] != C:
            raise ValueError("amp_prepare: need one int64 dataset index per agent row and rows of the transition's width")
        indices = indices.contiguous()
    dev = rms.mean.device
    agent_out = torch.empty((N, C), dtype=torch.float32, device=dev)
    expert_out = torch.empty((N, C), dtype=torch.float32, device=dev)
    workspace = torch.empty(max(int(_native.lib().cusrl_amp_prepare_workspace(N, C)), 1), dtype=torch.float64, device=dev)
    ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    check(_native.lib().cusrl_amp_prepare(
        ptr(state), ptr(next_state), 0 if state is None else state.shape[1], ptr(columns), K, ptr(agent_raw), ptr(dataset),
        ptr(indices), ptr(expert_raw), N, C, rms.mean.data_ptr(), rms.var.data_ptr(), rms.std.data_ptr(), rms._count.data_ptr(),
        float(rms.epsilon), -1.0 if rms.max_count is None else float(rms.max_count), -1.0 if rms.clamp is None else float(rms.clamp),
        agent_out.data_ptr(), expert_out.data_ptr(), workspace.data_ptr(), _stream()), "cusrl_amp_prepare")
    return agent_out, expert_out


def reward_shaping_(reward: torch.Tensor, scale: float, shift: float, lower: float | None, upper: float | None) -> torch.Tensor:
    """``reward.mul_(scale).add_(shift).clamp_(lower, upper)`` (reward.py:43-47) in place, one launch."""
    require_device(reward, "reward")
    if reward.dtype != torch.float32 or not reward.is_contiguous():
        raise TypeError("reward_shaping_: expected a contiguous float32 tensor")
    check(_native.lib().cusrl_reward_shaping(reward.data_ptr(), float(scale), float(shift), 0.0 if lower is None else float(lower),
                                             0.0 if upper is None else float(upper), int(lower is not None), int(upper is not None),
                                             reward.numel(), _stream()), "cusrl_reward_shaping")
    _modified_in_place(reward)
    return reward


def mse_loss_fwd_bwd(prediction: torch.Tensor, target: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """``(mean((prediction - target)^2), d loss / d prediction)`` from one pass (+ a one-block finalize)."""
    prediction, target = _f32(prediction, "prediction"), _f32(target, "target")
    if prediction.shape != target.shape or prediction.numel() == 0:
        raise ValueError("mse_loss_fwd_bwd: shapes differ or are empty")
    lib = _native.lib()
    n = prediction.numel()
    loss = torch.empty((), dtype=torch.float32, device=prediction.device)
    grad = torch.empty_like(prediction)
    partials = torch.empty(max(int(lib.cusrl_mse_loss_num_partials(n)), 1), dtype=torch.float64, device=prediction.device)
    check(lib.cusrl_mse_loss_fwd_bwd(prediction.data_ptr(), target.data_ptr(), n, loss.data_ptr(), grad.data_ptr(),
                                     partials.data_ptr(), _stream()), "cusrl_mse_loss_fwd_bwd")
    return loss, grad


def sumsq_fwd_bwd(x: torch.Tensor, loss_scale: float, grad_scale: float) -> tuple[torch.Tensor, torch.Tensor]:
    """``(loss_scale * sum(x^2), grad_scale * x)`` from one pass (AMP's gradient penalty and what it sends back)."""
    x = _f32(x, "x")
    lib = _native.lib()
    n = x.numel()
    loss = torch.empty((), dtype=torch.float32, device=x.device)
    grad = torch.empty_like(x)
    partials = torch.empty(max(int(lib.cusrl_mse_loss_num_partials(n)), 1), dtype=torch.float64, device=x.device)
    check(lib.cusrl_sumsq_fwd_bwd(x.data_ptr(), n, float(loss_scale), float(grad_scale), loss.data_ptr(), grad.data_ptr(),
                                  partials.data_ptr(), _stream()), "cusrl_sumsq_fwd_bwd")
    return loss, grad


def bce_pair_fwd_bwd(logit: torch.Tensor, weight: float) -> tuple[torch.Tensor, torch.Tensor]:
    """Discrimination loss of a joint ``[2N, 1]`` logit batch (agent rows first: target 0, expert rows: target 1) times
    ``weight``, and its gradient wrt the logits — one launch."""
    logit = _f32(logit, "logit")
    if logit.numel() % 2:
        raise ValueError("bce_pair_fwd_bwd: the joint batch holds as many expert as agent rows")
    loss = torch.empty((), dtype=torch.float32, device=logit.device)
    grad = torch.empty_like(logit)
    check(_native.lib().cusrl_bce_pair_fwd_bwd(logit.data_ptr(), logit.numel() // 2, float(weight), loss.data_ptr(), grad.data_ptr(),
                                               _stream()), "cusrl_bce_pair_fwd_bwd")
    return loss, grad


def accumulate_scalars_(accumulator: torch.Tensor, values: Sequence[torch.Tensor]) -> None:
    """``accumulator[i] += values[i]`` for 0-d fp32 device tensors — ONE launch per 32 values (pointer table by value)."""
    import ctypes

    if accumulator.dtype != torch.float32 or not accumulator.is_contiguous() or accumulator.numel() < len(values):
        raise ValueError("accumulate_scalars_: need a contiguous float32 accumulator with one slot per value")
    staged = [v if v.dtype == torch.float32 else v.float() for v in values]
    for start in range(0, len(staged), 32):
        chunk = staged[start:start + 32]
        table = (ctypes.c_void_p * len(chunk))(*[require_device(v, "value").data_ptr() for v in chunk])
        check(_native.lib().cusrl_accumulate_scalars(table, len(chunk), accumulator.data_ptr() + 4 * start, _stream()),
              "cusrl_accumulate_scalars")


def synthetic_env_step(seed: int, counter: torch.Tensor, num_envs: int, obs_dim: int, reward_dim: int, p_terminate: float,
                       p_truncate: float):
    """One step of the i.i.d. benchmark env as ONE launch (``cusrl_synthetic_env_step``): returns
    ``(next_observation [N, obs], reward [N, R], terminated [N, 1] bool, truncated [N, 1] bool, reset_rows [N, obs])``.
    ``counter``: int64[2] device tensor (zeros at construction) that the launch itself advances."""
    require_device(counter, "counter")
    dev = counter.device
    next_observation = torch.empty((num_envs, obs_dim), dtype=torch.float32, device=dev)
    reset_rows = torch.empty((num_envs, obs_dim), dtype=torch.float32, device=dev)
    reward = torch.empty((num_envs, reward_dim), dtype=torch.float32, device=dev)
    terminated = torch.empty((num_envs, 1), dtype=torch.bool, device=dev)
    truncated = torch.empty((num_envs, 1), dtype=torch.bool, device=dev)
    check(_native.lib().cusrl_synthetic_env_step(seed & 0xFFFFFFFFFFFFFFFF, counter.data_ptr(), num_envs, obs_dim, reward_dim,
                                                 float(p_terminate), float(p_truncate), next_observation.data_ptr(), reward.data_ptr(),
                                                 terminated.data_ptr(), truncated.data_ptr(), reset_rows.data_ptr(), _stream()),
          "cusrl_synthetic_env_step")
    return next_observation, reward, terminated, truncated, reset_rows
