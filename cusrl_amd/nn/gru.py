"""``torch.nn.GRU`` / ``torch.nn.LSTM`` evaluated as GEMMs + one HIP pass per time step (the recurrent backbones
cusrl/nn/module/rnn.py:21-120 wraps: GRU in config 4, LSTM by default in RecurrentPpoAgentFactory).

On ROCm ``nn.GRU`` is MIOpen's RNN, which spends a BPTT minibatch of config 4 in ~1 200 generic tensor kernels and a
7.6 ms bias-gradient reduction per layer and direction of differentiation (72 ms per minibatch step).  A GRU layer over a
padded ``[L, B, I]`` batch is

* ``gi = x W_ih^T + b_ih`` for all steps — ONE GEMM;
* per step ``gh = h W_hh^T`` (rocBLAS) and ONE pass over the gates (``cusrl_gru_gates_fwd``);
* backward: per step the gate pass in reverse (``cusrl_gru_gates_bwd``, overwriting the saved pre-activations with their
  gradients) and ``dh += d_gh W_hh``; then dW_ih / dW_hh as batched GEMMs over the L steps (L slabs: enough workgroups
  for 256 CUs where a single ``[3H, L*B] x [L*B, I]`` GEMM would not split its reduction), the bias gradients as column
  sums, dx as one GEMM.

Parameters stay those of ``nn.GRU`` (same names, same state dict).  ``lengths`` gives the result a PackedSequence would
give — the state stops at each sequence's last valid step, ended positions emit zeros — by sorting the batch by length
and shrinking the per-step launches to the sequences still running (one small host read of the step sizes; the gate
kernels also take device-side ``lengths`` for an unsorted batch).
"""

from __future__ import annotations

import os

import torch
from torch import Tensor

from cusrl_amd import ops

__all__ = ["gru_forward", "gru_supported", "lstm_forward", "rnn_forward"]


def _gemm_rows(n: int, B: int) -> int:
    """Row count of a time step's recurrent GEMM: the ``n`` running sequences rounded up to a multiple of 256 (at most
    the batch).  The extra rows belong to ended sequences — their state is frozen and finite, the gate pass ignores their
    projections, and in the backward pass their gate gradients are zeros — so the products over them change nothing; the
    point is that the GEMM shapes no longer depend on where episodes happened to end, which lets the measured kernel
    selection (cusrl_amd/tuned_gemms_gfx950.csv) cover them: rocBLAS's default choice for ``[~5000, 256] x [256, 768]`` runs
    at half the speed of its best kernel."""
    return min(B, (n + 255) // 256 * 256)


class _GruLayer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, h0: Tensor, w_ih: Tensor, w_hh: Tensor, b_ih: Tensor | None, b_hh: Tensor | None,
                lengths: Tensor | None, sizes: list[int] | None):
        L, B, I = x.shape
        H = w_hh.shape[1]
        flat = x.reshape(L * B, I)
        gi = (torch.mm(flat, w_ih.t()) if b_ih is None else torch.addmm(b_ih, flat, w_ih.t())).view(L, B, 3 * H)
        keep = any(ctx.needs_input_grad[:6])
        gh = torch.empty((L if keep else 1, B, 3 * H), dtype=x.dtype, device=x.device)
        # the outputs live behind one extra leading slab that holds h0: `states[:-1]` then IS the array of the states that
        # entered every step — what the recurrent weight gradient pairs with d_gh — without a cat of 136 MB per layer
        states = _new_output(x, L + 1, B, H, None if sizes is None else [B] + sizes)
        states[0].copy_(h0)
        out = states[1:]
        h = h0.clone(memory_format=torch.contiguous_format)
        w_hh_t = w_hh.t()
        for t in range(L):
            n = B if sizes is None else sizes[t]  # sorted by length: the sequences still running are the first n rows
            if n == 0:
                break
            gh_t = gh[t if keep else 0]
            rows = _gemm_rows(n, B)
            torch.mm(h[:rows], w_hh_t, out=gh_t[:rows])
            ops.gru_gates_forward(gi[t, :n], gh_t[:n], b_hh, h[:n], out[t, :n], lengths, t)
        if keep:
            ctx.save_for_backward(x, h0, w_ih, w_hh, b_hh, lengths, out)
            ctx.gi, ctx.gh, ctx.has_b_ih, ctx.sizes, ctx.states = gi, gh, b_ih is not None, sizes, states
        return out, h

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out: Tensor | None, d_last: Tensor | None):
        x, h0, w_ih, w_hh, b_hh, lengths, out = ctx.saved_tensors
        gi, gh = ctx.gi, ctx.gh
        if gi is None:
            raise RuntimeError("the fused GRU layer keeps its pre-activations for ONE backward pass (they are overwritten "
                               "with their gradients); backward(retain_graph=True) followed by a second pass is not supported")
        ctx.gi = ctx.gh = None
        L, B, _ = x.shape
        H = w_hh.shape[1]
        dh = torch.zeros((B, H), dtype=x.dtype, device=x.device) if d_last is None else d_last.contiguous().clone()
        if d_out is not None:
            d_out = d_out.contiguous()
        h0 = h0.contiguous()
        sizes = ctx.sizes
        # Length-sorted batch with the sorted lengths at hand: the gate pass runs over ALL rows and zeroes the gate
        # gradients of the ended sequences itself (its `lengths` branch) — instead of two fill launches per step for the
        # tails (config 4: 3 840 of them per iteration, 6.8 % of the device time).
        in_kernel_tails = sizes is not None and lengths is not None
        need = ctx.needs_input_grad
        # bias gradients folded into the gate pass: every block leaves the column sums of the gate gradients it writes
        # (ops.gru_gates_backward, `bias_partials`), summed ONCE per layer below — instead of two column-sum passes over the
        # [L * B, 3H] gradient arrays (config 4: 14 ms per iteration of re-reading 408 MB arrays at the HBM roofline)
        want_bias = (ctx.has_b_ih and need[4]) or (b_hh is not None and need[5])
        bias_rows = ops.gru_bias_partial_rows(B)
        bias_partials = None
        if want_bias and ops.gru_bias_partials_supported(H, gi, gh, b_hh, h0, d_out, dh) and out.data_ptr() % 16 == 0:
            # partial rows no launch writes (row subsets of a length-sorted batch without device-side lengths, skipped
            # steps) must read as zero
            full = sizes is None or in_kernel_tails and sizes[-1] > 0
            bias_partials = (torch.empty if full else torch.zeros)((L, bias_rows, 4 * H), dtype=x.dtype, device=x.device)
        for t in range(L - 1, -1, -1):
            n = B if sizes is None else sizes[t]
            h_prev = h0 if t == 0 else out[t - 1]
            if in_kernel_tails and n > 0:
                ops.gru_gates_backward(gi[t], gh[t], b_hh, h_prev, None if d_out is None else d_out[t], dh, lengths, t,
                                       None if bias_partials is None else bias_partials[t])
                rows = _gemm_rows(n, B)
                dh[:rows].addmm_(gh[t, :rows], w_hh)  # + d_gh_t @ W_hh (zero rows for the ended sequences)
                continue
            if n < B:  # ended sequences: no gate gradients (their rows still hold the forward's projections)
                gi[t, n:].zero_()
                gh[t, n:].zero_()
            if n == 0:
                continue
            partial = None if bias_partials is None else bias_partials[t, : ops.gru_bias_partial_rows(n)]
            ops.gru_gates_backward(gi[t, :n], gh[t, :n], b_hh, h_prev[:n], None if d_out is None else d_out[t, :n],
                                   dh[:n], lengths, t, partial)
            dh[:n].addmm_(gh[t, :n], w_hh)  # + d_gh_t @ W_hh
        d_x = d_w_ih = d_w_hh = d_b_ih = d_b_hh = None
        if need[0]:
            d_x = torch.mm(gi.view(L * B, 3 * H), w_ih).view(x.shape)
        if need[2]:
            d_w_ih = _sum_slabs(w_ih, torch.bmm(gi.transpose(1, 2), x)) if L > 1 else torch.mm(gi[0].t(), x[0])
        if need[3]:
            # slab t pairs d_gh[t] with the state that entered step t: h0 for t = 0, out[t - 1] after that = states[:-1]
            if ctx.states is None:
                raise RuntimeError("the fused GRU layer keeps its states for ONE backward pass (they are released with it): "
                                   "a second backward through the same graph needs a fresh forward")
            h_prev = ctx.states[:-1]
            ctx.states = None
            d_w_hh = _sum_slabs(w_hh, torch.bmm(gh.transpose(1, 2), h_prev)) if L > 1 else torch.mm(gh[0].t(), h_prev[0])
        if bias_partials is not None:
            sums = _column_sums(bias_partials.view(L * bias_rows, 4 * H))  # {sum d_r, sum d_z, sum d_n, sum d_q}
            if ctx.has_b_ih and need[4]:
                d_b_ih = sums[: 3 * H]
            if b_hh is not None and need[5]:
                d_b_hh = torch.cat((sums[: 2 * H], sums[3 * H:]))
        else:
            if ctx.has_b_ih and need[4]:
                d_b_ih = _column_sums(gi.view(L * B, 3 * H))
            if b_hh is not None and need[5]:
                d_b_hh = _column_sums(gh.view(L * B, 3 * H))
        return d_x, (dh if need[1] else None), d_w_ih, d_w_hh, d_b_ih, d_b_hh, None, None


def _sum_slabs(weight: Tensor, slabs: Tensor) -> Tensor | None:
    """A weight gradient that is still ``L`` per-step slabs ``[L, out, in]``: handed to the flat-gradient assembly when
    one is collecting (it sums slabs straight into the parameter's slot — ``nn/module.py`` does the same for the MLP's
    split-batch weight gradients — and autograd then gets no gradient for this parameter); summed here otherwise."""
    from cusrl_amd.nn import module as nn_module

    sink = nn_module._split_grad_sink
    key = weight.data_ptr()
    if sink is not None and weight.is_leaf and key not in sink:
        sink[key] = slabs.view(slabs.shape[0], -1)
        return None
    return ops.sum_slabs(slabs) if slabs.is_cuda and slabs.dtype == torch.float32 else slabs.sum(0)


def _new_output(x: Tensor, L: int, B: int, H: int, sizes: list[int] | None) -> Tensor:
    """Output buffer of a layer: steps / rows no launch will touch (ended sequences of a length-sorted batch) read as zero."""
    if sizes is not None and sizes[-1] < B:
        return torch.zeros((L, B, H), dtype=x.dtype, device=x.device)
    return torch.empty((L, B, H), dtype=x.dtype, device=x.device)


class _LstmLayer(torch.autograd.Function):
    """The same decomposition for ``nn.LSTM`` (the default core of RecurrentPpoAgentFactory).  Both biases are additive, so
    the layer keeps ONE ``[L, B, 4H]`` array (the summed pre-activations, later their gradient) plus the cell states."""

    @staticmethod
    def forward(ctx, x: Tensor, h0: Tensor, c0: Tensor, w_ih: Tensor, w_hh: Tensor, b_ih: Tensor | None,
                b_hh: Tensor | None, lengths: Tensor | None, sizes: list[int] | None):
        L, B, I = x.shape
        H = w_hh.shape[1]
        flat = x.reshape(L * B, I)
        gi = (torch.mm(flat, w_ih.t()) if b_ih is None else torch.addmm(b_ih, flat, w_ih.t())).view(L, B, 4 * H)
        keep = any(ctx.needs_input_grad[:7])
        gh = torch.empty((B, 4 * H), dtype=x.dtype, device=x.device)
        out = _new_output(x, L, B, H, sizes)
        cells = torch.empty((L, B, H), dtype=x.dtype, device=x.device) if keep else None
        h, c = h0.clone(memory_format=torch.contiguous_format), c0.clone(memory_format=torch.contiguous_format)
        w_hh_t = w_hh.t()
        for t in range(L):
            n = B if sizes is None else sizes[t]
            if n == 0:
                break
            torch.mm(h[:n], w_hh_t, out=gh[:n])
            ops.lstm_gates_forward(gi[t, :n], gh[:n], b_hh, h[:n], c[:n], out[t, :n], None if cells is None else cells[t, :n],
                                   lengths, t)
        if keep:
            ctx.save_for_backward(x, h0, c0, w_ih, w_hh, lengths, out, cells)
            ctx.pre, ctx.has_b_ih, ctx.has_b_hh, ctx.sizes = gi, b_ih is not None, b_hh is not None, sizes
        return out, h, c

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out: Tensor | None, d_last_h: Tensor | None, d_last_c: Tensor | None):
        x, h0, c0, w_ih, w_hh, lengths, out, cells = ctx.saved_tensors
        pre = ctx.pre
        if pre is None:
            raise RuntimeError("the fused LSTM layer keeps its pre-activations for ONE backward pass (they are overwritten "
                               "with their gradients); backward(retain_graph=True) followed by a second pass is not supported")
        ctx.pre = None
        L, B, _ = x.shape
        H = w_hh.shape[1]
        zeros = lambda: torch.zeros((B, H), dtype=x.dtype, device=x.device)  # noqa: E731
        dh = zeros() if d_last_h is None else d_last_h.contiguous().clone()
        dc = zeros() if d_last_c is None else d_last_c.contiguous().clone()
        if d_out is not None:
            d_out = d_out.contiguous()
        h0, c0 = h0.contiguous(), c0.contiguous()
        sizes = ctx.sizes
        for t in range(L - 1, -1, -1):
            n = B if sizes is None else sizes[t]
            if n < B:
                pre[t, n:].zero_()
            if n == 0:
                continue
            c_prev = c0 if t == 0 else cells[t - 1]
            ops.lstm_gates_backward(pre[t, :n], c_prev[:n], cells[t, :n], None if d_out is None else d_out[t, :n],
                                    dh[:n], dc[:n], lengths, t)
            dh[:n].addmm_(pre[t, :n], w_hh)  # + d_pre_t @ W_hh (dh holds what bypassed the step: ended sequences only)
        need = ctx.needs_input_grad
        d_x = d_w_ih = d_w_hh = d_b = None
        if need[0]:
            d_x = torch.mm(pre.view(L * B, 4 * H), w_ih).view(x.shape)
        if need[3]:
            d_w_ih = ops.sum_slabs(torch.bmm(pre.transpose(1, 2), x)) if L > 1 else torch.mm(pre[0].t(), x[0])
        if need[4]:
            h_prev = torch.cat([h0.unsqueeze(0), out[:-1]]) if L > 1 else h0.unsqueeze(0)
            d_w_hh = ops.sum_slabs(torch.bmm(pre.transpose(1, 2), h_prev)) if L > 1 else torch.mm(pre[0].t(), h_prev[0])
        if (ctx.has_b_ih and need[5]) or (ctx.has_b_hh and need[6]):
            d_b = _column_sums(pre.view(L * B, 4 * H))
        return (d_x, dh if need[1] else None, dc if need[2] else None, d_w_ih, d_w_hh,
                d_b if ctx.has_b_ih and need[5] else None, d_b if ctx.has_b_hh and need[6] else None, None, None)


class _RnnLayer(torch.autograd.Function):
    """``nn.RNN`` (tanh | relu): the output is the only saved state; the projection buffer becomes the gradient buffer."""

    @staticmethod
    def forward(ctx, x: Tensor, h0: Tensor, w_ih: Tensor, w_hh: Tensor, b_ih: Tensor | None, b_hh: Tensor | None,
                lengths: Tensor | None, relu: bool, sizes: list[int] | None):
        L, B, I = x.shape
        H = w_hh.shape[1]
        flat = x.reshape(L * B, I)
        gi = (torch.mm(flat, w_ih.t()) if b_ih is None else torch.addmm(b_ih, flat, w_ih.t())).view(L, B, H)
        gh = torch.empty((B, H), dtype=x.dtype, device=x.device)
        out = _new_output(x, L, B, H, sizes)
        h = h0.clone(memory_format=torch.contiguous_format)
        w_hh_t = w_hh.t()
        for t in range(L):
            n = B if sizes is None else sizes[t]
            if n == 0:
                break
            torch.mm(h[:n], w_hh_t, out=gh[:n])
            ops.rnn_cell_forward(gi[t, :n], gh[:n], b_hh, h[:n], out[t, :n], lengths, t, relu)
        if any(ctx.needs_input_grad[:6]):
            ctx.save_for_backward(x, h0, w_ih, w_hh, lengths, out)
            ctx.scratch, ctx.relu, ctx.has_b_ih, ctx.has_b_hh, ctx.sizes = gi, relu, b_ih is not None, b_hh is not None, sizes
        return out, h

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out: Tensor | None, d_last: Tensor | None):
        x, h0, w_ih, w_hh, lengths, out = ctx.saved_tensors
        d_pre = ctx.scratch  # the projections are no longer needed: reuse their memory for the gradients
        L, B, _ = x.shape
        H = w_hh.shape[1]
        dh = torch.zeros((B, H), dtype=x.dtype, device=x.device) if d_last is None else d_last.contiguous().clone()
        if d_out is not None:
            d_out = d_out.contiguous()
        h0 = h0.contiguous()
        sizes = ctx.sizes
        for t in range(L - 1, -1, -1):
            n = B if sizes is None else sizes[t]
            if n < B:
                d_pre[t, n:].zero_()
            if n == 0:
                continue
            ops.rnn_cell_backward(d_pre[t, :n], out[t, :n], None if d_out is None else d_out[t, :n], dh[:n], lengths, t, ctx.relu)
            dh[:n].addmm_(d_pre[t, :n], w_hh)
        need = ctx.needs_input_grad
        d_x = d_w_ih = d_w_hh = d_b = None
        if need[0]:
            d_x = torch.mm(d_pre.view(L * B, H), w_ih).view(x.shape)
        if need[2]:
            d_w_ih = ops.sum_slabs(torch.bmm(d_pre.transpose(1, 2), x)) if L > 1 else torch.mm(d_pre[0].t(), x[0])
        if need[3]:
            h_prev = torch.cat([h0.unsqueeze(0), out[:-1]]) if L > 1 else h0.unsqueeze(0)
            d_w_hh = ops.sum_slabs(torch.bmm(d_pre.transpose(1, 2), h_prev)) if L > 1 else torch.mm(d_pre[0].t(), h_prev[0])
        if (ctx.has_b_ih and need[4]) or (ctx.has_b_hh and need[5]):
            d_b = _column_sums(d_pre.view(L * B, H))
        return (d_x, dh if need[1] else None, d_w_ih, d_w_hh, d_b if ctx.has_b_ih and need[4] else None,
                d_b if ctx.has_b_hh and need[5] else None, None, None, None)


def _column_sums(matrix: Tensor) -> Tensor:
    if matrix.is_cuda and matrix.dtype == torch.float32:
        return ops.relu_backward_bias(matrix.contiguous(), None)[1]  # the one-pass column-sum kernel of the MLP's bias gradients
    return matrix.sum(0)


class _LengthPlan:
    """Sequences sorted by decreasing length (what a PackedSequence does): at step t the ones still running are the first
    ``sizes[t]`` rows, so the per-step GEMM and gate pass shrink with the batch instead of computing padded rows — a
    done-split BPTT minibatch of config 4 carries 5 500 sequences for 4 096 x 24 valid steps, i.e. 26 % padding.  Costs
    one stable sort of the lengths and ONE host read (the L step sizes); inputs / outputs are permuted by index_select."""

    def __init__(self, lengths: Tensor, steps: int):
        ordered, self.order = torch.sort(lengths, descending=True, stable=True)
        self.ordered = ordered  # the lengths in sorted order (the gate passes' `lengths` argument)
        self.inverse = torch.empty_like(self.order)
        self.inverse[self.order] = torch.arange(self.order.numel(), device=self.order.device)
        running = ordered.unsqueeze(0) > torch.arange(steps, device=lengths.device).unsqueeze(1)
        self.sizes: list[int] = running.sum(1).tolist()

    def sort(self, tensor: Tensor, dim: int) -> Tensor:
        if _permutable(tensor, dim):
            return _PermuteRows.apply(tensor, self.order, self.inverse)
        return tensor.index_select(dim, self.order)

    def unsort(self, tensor: Tensor, dim: int) -> Tensor:
        if _permutable(tensor, dim):
            return _PermuteRows.apply(tensor, self.inverse, self.order)
        return tensor.index_select(dim, self.inverse)


def _permutable(tensor: Tensor, dim: int) -> bool:
    return dim == 1 and tensor.dim() == 3 and tensor.is_cuda and tensor.dtype == torch.float32


class _PermuteRows(torch.autograd.Function):
    """``tensor[:, index]`` for a PERMUTATION ``index`` of the batch axis of a ``[L, B, C]`` tensor through the row-gather
    kernel (``cusrl_gather_rows``, ``[:, idx]`` form); the backward of a permutation is the gather with its inverse — no
    index_add.  (torch: a scatter/gather kernel forward and ``indexFuncLargeIndex`` backward, 12 ms per iteration of config 4.)"""

    @staticmethod
    def forward(ctx, tensor: Tensor, index: Tensor, inverse: Tensor):
        ctx.save_for_backward(index, inverse)
        tensor = tensor.contiguous()
        return ops.gather_rows([tensor], index, tensor.shape[0], tensor.shape[1], temporal=True)[0]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad: Tensor):
        index, inverse = ctx.saved_tensors
        grad = grad.contiguous()
        return ops.gather_rows([grad], inverse, grad.shape[0], grad.shape[1], temporal=True)[0], None, None


def _plan(lengths: Tensor | None, input: Tensor) -> _LengthPlan | None:
    if lengths is None:
        return None
    lengths = lengths.to(device=input.device, dtype=torch.int64)
    if lengths.numel() != input.shape[1]:
        raise ValueError(f"'lengths' has {lengths.numel()} entries for a batch of {input.shape[1]} sequences")
    return _LengthPlan(lengths, input.shape[0])


def gru_supported(module: torch.nn.RNNBase, input) -> bool:
    """fp32 device tensors through a plain (uni-directional, time-major, dropout-free at this call) ``nn.GRU`` / ``nn.LSTM``."""
    if os.environ.get("CUSRL_FUSED_RNN", "1") == "0":  # escape hatch / A-B switch: MIOpen's RNN for everything
        return False
    return (isinstance(input, Tensor) and input.is_cuda and input.dtype == torch.float32 and input.dim() == 3
            and not module.bidirectional and not module.batch_first and getattr(module, "proj_size", 0) == 0
            and (module.dropout == 0.0 or not module.training or module.num_layers == 1)
            and module.weight_ih_l0.dtype == torch.float32 and not torch.is_autocast_enabled("cuda"))


def gru_forward(module: torch.nn.GRU, input: Tensor, h0: Tensor | None, lengths: Tensor | None = None):
    """``module(input, h0)`` for ``input [L, B, I]`` and ``h0 [layers, B, H]`` (zeros if None) -> ``(output [L, B, H],
    h_n [layers, B, H])``; with ``lengths`` (int64 [B] on the device) as for the packed form of the same batch."""
    L, B, _ = input.shape
    H, layers = module.hidden_size, module.num_layers
    if h0 is None:
        h0 = torch.zeros((layers, B, H), dtype=input.dtype, device=input.device)
    plan = _plan(lengths, input)
    sizes = None if plan is None else plan.sizes
    if plan is not None:
        input, h0 = plan.sort(input, 1), plan.sort(h0, 1)
    x, finals = input.contiguous(), []
    for layer in range(layers):
        w_ih, w_hh = getattr(module, f"weight_ih_l{layer}"), getattr(module, f"weight_hh_l{layer}")
        b_ih = getattr(module, f"bias_ih_l{layer}") if module.bias else None
        b_hh = getattr(module, f"bias_hh_l{layer}") if module.bias else None
        x, last = _GruLayer.apply(x, h0[layer].contiguous(), w_ih, w_hh, b_ih, b_hh, None if plan is None else plan.ordered, sizes)
        finals.append(last)
    last = torch.stack(finals)
    return (x, last) if plan is None else (plan.unsort(x, 1), plan.unsort(last, 1))


def lstm_forward(module: torch.nn.LSTM, input: Tensor, state: tuple[Tensor, Tensor] | None, lengths: Tensor | None = None):
    """``module(input, (h0, c0))`` for ``input [L, B, I]`` and states ``[layers, B, H]`` (zeros if None) ->
    ``(output, (h_n, c_n))``; ``lengths`` as in :func:`gru_forward`."""
    L, B, _ = input.shape
    H, layers = module.hidden_size, module.num_layers
    if state is None:
        h0 = c0 = torch.zeros((layers, B, H), dtype=input.dtype, device=input.device)
    else:
        h0, c0 = state
    plan = _plan(lengths, input)
    sizes = None if plan is None else plan.sizes
    if plan is not None:
        input, h0, c0 = plan.sort(input, 1), plan.sort(h0, 1), plan.sort(c0, 1)
    x, last_h, last_c = input.contiguous(), [], []
    for layer in range(layers):
        w_ih, w_hh = getattr(module, f"weight_ih_l{layer}"), getattr(module, f"weight_hh_l{layer}")
        b_ih = getattr(module, f"bias_ih_l{layer}") if module.bias else None
        b_hh = getattr(module, f"bias_hh_l{layer}") if module.bias else None
        x, h, c = _LstmLayer.apply(x, h0[layer].contiguous(), c0[layer].contiguous(), w_ih, w_hh, b_ih, b_hh, None, sizes)
        last_h.append(h), last_c.append(c)
    hn, cn = torch.stack(last_h), torch.stack(last_c)
    return (x, (hn, cn)) if plan is None else (plan.unsort(x, 1), (plan.unsort(hn, 1), plan.unsort(cn, 1)))


def rnn_forward(module: torch.nn.RNN, input: Tensor, h0: Tensor | None, lengths: Tensor | None = None):
    """``module(input, h0)`` for a tanh / relu ``nn.RNN``; arguments and results as :func:`gru_forward`."""
    L, B, _ = input.shape
    H, layers = module.hidden_size, module.num_layers
    if h0 is None:
        h0 = torch.zeros((layers, B, H), dtype=input.dtype, device=input.device)
    plan = _plan(lengths, input)
    sizes = None if plan is None else plan.sizes
    if plan is not None:
        input, h0 = plan.sort(input, 1), plan.sort(h0, 1)
    relu = module.nonlinearity == "relu"
    x, finals = input.contiguous(), []
    for layer in range(layers):
        w_ih, w_hh = getattr(module, f"weight_ih_l{layer}"), getattr(module, f"weight_hh_l{layer}")
        b_ih = getattr(module, f"bias_ih_l{layer}") if module.bias else None
        b_hh = getattr(module, f"bias_hh_l{layer}") if module.bias else None
        x, last = _RnnLayer.apply(x, h0[layer].contiguous(), w_ih, w_hh, b_ih, b_hh, None, relu, sizes)
        finals.append(last)
    last = torch.stack(finals)
    return (x, last) if plan is None else (plan.unsort(x, 1), plan.unsort(last, 1))
