"""Done-split sequence packing for BPTT minibatches (counterpart of cusrl/nn/utils/recurrent.py:63-272).

A temporal minibatch ``[L, N, ...]`` contains episode boundaries (``done``).  Before the RNN sees it, every env's
column is cut at its boundaries into separate sequences that each start from a zero (or the stored) memory:
``split_and_pad_sequences`` lays them out as ``[L, Ns, ...]`` (env-major sequence order, zero padded) with a validity
``mask [L, Ns]``, ``scatter_memory`` places each env's stored memory at its first sequence, and
``unpad_and_merge_sequences`` maps the RNN output back to ``[L, N, ...]``.

On MI355X the layout is computed by two small HIP launches (``cusrl_sequence_count`` / ``cusrl_sequence_layout``) with
ONE host read (the number of sequences, needed to size the padded tensors — the reference synchronises several times:
``nonzero``, ``.item()``, boolean-mask scatters), and the data moves through the same row scatter / gather kernels as
the rest of the path (16-byte lanes, one launch per tensor).
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Any

import torch
from torch import Tensor

from cusrl_amd import _native, ops
from cusrl_amd._native import check
from cusrl_amd.utils.nest import map_nested

__all__ = [
    "SequenceLayout",
    "compute_cumulative_sequence_lengths",
    "compute_cumulative_timesteps",
    "compute_reverse_cumulative_timesteps",
    "compute_sequence_indices",
    "compute_sequence_layout",
    "compute_sequence_lengths",
    "cumulate_sequence_lengths",
    "gather_memory",
    "scatter_memory",
    "select_initial_memory",
    "split_and_pad_sequences",
    "unpad_and_merge_sequences",
]


@dataclass
class SequenceLayout:
    """Where every slot ``(t, n)`` of an ``[L, N]`` batch lives inside the padded ``[L, Ns]`` layout."""

    length: int
    num_envs: int
    num_sequences: int
    dest: Tensor        # int64 [L * N]: pos * Ns + seq of slot t * N + n
    first_seq: Tensor   # int64 [N]: index of env n's first sequence
    mask: Tensor        # bool [L, Ns]
    lengths: Tensor     # int64 [Ns]: valid steps of every sequence
    last_seq: Tensor    # int64 [N]: index of the sequence still open at the end of env n's column
    done_last: Tensor   # bool [N]: done[-1] (whether that sequence ended exactly at the last step)


class _PackRows(torch.autograd.Function):
    """``padded.flatten(0, 1)[dest[k]] = x.flatten(0, 1)[k]`` (zeros elsewhere); backward gathers the same rows."""

    @staticmethod
    def forward(ctx, x, dest, length, num_sequences):
        ctx.save_for_backward(dest)
        ctx.shape = x.shape
        x = x.contiguous()
        padded = torch.zeros((length, num_sequences) + tuple(x.shape[2:]), dtype=x.dtype, device=x.device)
        ops.scatter_rows(x.flatten(0, 1), dest, padded)
        return padded

    @staticmethod
    def backward(ctx, grad_padded):
        (dest,) = ctx.saved_tensors
        grad_padded = grad_padded.contiguous()
        (rows,) = ops.gather_rows([grad_padded], dest, grad_padded.size(0), grad_padded.size(1))
        return rows.view(ctx.shape), None, None, None


class _UnpackRows(torch.autograd.Function):
    """``out.flatten(0, 1)[k] = padded.flatten(0, 1)[dest[k]]``; backward scatters into a zero padded tensor."""

    @staticmethod
    def forward(ctx, padded, dest, length, num_envs):
        ctx.save_for_backward(dest)
        ctx.shape = padded.shape
        padded = padded.contiguous()
        (rows,) = ops.gather_rows([padded], dest, padded.size(0), padded.size(1))
        return rows.view((length, num_envs) + tuple(padded.shape[2:]))

    @staticmethod
    def backward(ctx, grad_out):
        (dest,) = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        grad_padded = torch.zeros(ctx.shape, dtype=grad_out.dtype, device=grad_out.device)
        ops.scatter_rows(grad_out.flatten(0, 1), dest, grad_padded)
        return grad_padded, None, None, None


def _check_done(done: Tensor) -> Tensor:
    if done.dim() != 3 or done.size(-1) != 1:
        raise ValueError(f"'done' must be a 3D tensor with a last dimension of 1; got shape {done.shape}")
    ops.require_device(done, "done")
    if done.dtype not in (torch.bool, torch.uint8):
        raise TypeError(f"'done' must have dtype bool, got {done.dtype}")
    return done.contiguous()


def compute_sequence_layout(done: Tensor) -> SequenceLayout:
    done = _check_done(done)
    L, N = done.shape[:2]
    lib = _native.lib()
    dev = done.device
    stream = torch.cuda.current_stream().cuda_stream
    env_prefix = torch.empty(N, dtype=torch.int32, device=dev)
    block_totals = torch.empty(int(lib.cusrl_sequence_blocks(N)), dtype=torch.int32, device=dev)
    total = torch.empty(1, dtype=torch.int32, device=dev)
    check(lib.cusrl_sequence_count(done.data_ptr(), L, N, env_prefix.data_ptr(), block_totals.data_ptr(), total.data_ptr(), stream),
          "cusrl_sequence_count")
    num_sequences = int(total.item())  # the one host read: sizes the padded tensors
    dest = torch.empty(L * N, dtype=torch.int64, device=dev)
    first_seq = torch.empty(N, dtype=torch.int64, device=dev)
    mask = torch.zeros(L, num_sequences, dtype=torch.bool, device=dev)
    lengths = torch.empty(num_sequences, dtype=torch.int64, device=dev)
    last_seq = torch.empty(N, dtype=torch.int64, device=dev)
    check(lib.cusrl_sequence_layout(done.data_ptr(), L, N, env_prefix.data_ptr(), block_totals.data_ptr(), num_sequences,
                                    dest.data_ptr(), first_seq.data_ptr(), mask.data_ptr(), lengths.data_ptr(),
                                    last_seq.data_ptr(), stream), "cusrl_sequence_layout")
    return SequenceLayout(L, N, num_sequences, dest, first_seq, mask, lengths, last_seq, done[-1].reshape(N))


def split_and_pad_sequences(compact_sequences: Tensor, done: Tensor, layout: SequenceLayout | None = None) -> tuple[Tensor, Tensor]:
    """``[L, N, ...]`` -> (``[L, Ns, ...]`` zero padded, ``mask [L, Ns]``)."""
    if compact_sequences.dim() < 3:
        raise ValueError(f"'compact_sequences' must be at least 3D; got shape {compact_sequences.shape}")
    layout = layout or compute_sequence_layout(done)
    padded = _PackRows.apply(compact_sequences, layout.dest, layout.length, layout.num_sequences)
    layout.mask._cusrl_layout = layout  # lets unpad_and_merge_sequences(padded, mask) reuse the layout
    return padded, layout.mask


def unpad_and_merge_sequences(padded_sequences: Tensor, mask: Tensor | SequenceLayout, original_sequence_len: int | None = None) -> Tensor:
    """Inverse of :func:`split_and_pad_sequences`: ``[L, Ns, ...]`` -> ``[L, N, ...]``."""
    layout = mask if isinstance(mask, SequenceLayout) else getattr(mask, "_cusrl_layout", None)
    if layout is None:
        # a bare mask: the k-th valid (seq, pos) in row-major order is slot (n, t) = divmod(k, L) — ordered compaction
        L = original_sequence_len or padded_sequences.size(0)
        flat, count = ops.compact_flags(mask.transpose(0, 1).contiguous())
        valid = int(count.item())
        seq, pos = flat[:valid] // mask.size(0), flat[:valid] % mask.size(0)
        dest_env_major = pos * mask.size(1) + seq                  # indexed by n * L + t
        dest = dest_env_major.view(-1, L).transpose(0, 1).reshape(-1)
        num_envs = valid // L
    else:
        dest, num_envs, L = layout.dest, layout.num_envs, layout.length
    return _UnpackRows.apply(padded_sequences, dest, L, num_envs)


def scatter_memory(memory: Any, done: Tensor, layout: SequenceLayout | None = None) -> Any:
    """``[N, ...]`` stored memories -> ``[Ns, ...]``: each env's memory at its first sequence, zeros elsewhere."""
    if memory is None:
        return None
    layout = layout or compute_sequence_layout(done)

    def scatter(mem: Tensor) -> Tensor:
        result = mem.new_zeros((layout.num_sequences,) + tuple(mem.shape[1:]))
        ops.scatter_rows(mem.contiguous(), layout.first_seq, result)
        return result

    return map_nested(scatter, memory)


def select_initial_memory(memory: Any, expected_shape) -> Any:
    """Sequence-aligned memories ``[L, N, H]`` -> the memory at the first step."""
    if memory is None:
        return None
    return map_nested(lambda mem: mem[0] if mem.shape[:-1] == tuple(expected_shape) else mem, memory)


def compute_sequence_lengths(done: Tensor) -> Tensor:
    """Length of every sequence, env-major order (recurrent.py:63-92) — written by the layout kernel."""
    return compute_sequence_layout(done).lengths


def cumulate_sequence_lengths(sequence_lens: Tensor) -> Tensor:
    out = sequence_lens.new_zeros(sequence_lens.size(0) + 1)
    out[1:] = sequence_lens.cumsum(dim=0)
    return out


def compute_cumulative_sequence_lengths(done: Tensor) -> Tensor:
    return cumulate_sequence_lengths(compute_sequence_lengths(done))


def compute_cumulative_timesteps(done: Tensor, layout: SequenceLayout | None = None) -> Tensor:
    """``[L, N, 1]`` int64: position of every slot inside its own sequence, counted from the sequence's first step
    (recurrent.py:28-32) — the ``pos`` part of the layout kernel's padded-row index, no scan."""
    layout = layout or compute_sequence_layout(done)
    return (layout.dest // layout.num_sequences).view(layout.length, layout.num_envs, 1)


def compute_reverse_cumulative_timesteps(done: Tensor, layout: SequenceLayout | None = None) -> Tensor:
    """``[L, N, 1]`` int64: steps left until the slot's sequence ends (recurrent.py:95-99)."""
    layout = layout or compute_sequence_layout(done)
    position, sequence = layout.dest // layout.num_sequences, layout.dest % layout.num_sequences
    return (layout.lengths[sequence] - 1 - position).view(layout.length, layout.num_envs, 1)


def gather_memory(memory: Any, done: Tensor, layout: SequenceLayout | None = None) -> Any:
    """Inverse of :func:`scatter_memory` (recurrent.py:124-157): ``[Ns, ...]`` per-sequence states -> ``[N, ...]``, each env
    getting the state of the sequence still open at the end of its column, cleared where it ended exactly at the last
    step.  One HIP launch per tensor (``cusrl_gather_memory``)."""
    if memory is None:
        return None
    layout = layout or compute_sequence_layout(done)

    def gather(mem: Tensor) -> Tensor:
        ops.require_device(mem, "memory")
        mem = mem.contiguous()
        if mem.shape[0] != layout.num_sequences:
            raise ValueError(f"gather_memory: expected {layout.num_sequences} sequences, got {mem.shape[0]}")
        result = mem.new_empty((layout.num_envs,) + tuple(mem.shape[1:]))
        row_bytes = mem.element_size() * (mem.numel() // max(mem.shape[0], 1))
        check(_native.lib().cusrl_gather_memory(mem.data_ptr(), layout.last_seq.data_ptr(), layout.done_last.data_ptr(),
                                                result.data_ptr(), layout.num_envs, row_bytes,
                                                torch.cuda.current_stream().cuda_stream), "cusrl_gather_memory")
        return result

    return map_nested(gather, memory)


def compute_sequence_indices(done: Tensor) -> Tensor:
    """``[N + 1]`` cumulative number of sequences before each env (recurrent.py:35-60)."""
    layout = compute_sequence_layout(done)
    out = torch.empty(layout.num_envs + 1, dtype=torch.int64, device=done.device)
    out[:-1] = layout.first_seq
    out[-1] = layout.num_sequences
    return out
