"""torch.nn building blocks of the actor-critic (counterparts of cusrl/nn/module/{module,mlp}.py and
cusrl/nn/layer/linear.py).  Their dense contractions stay on rocBLAS / hipBLASLt MFMA kernels (BASELINE.json north_star);
what is hand-written here is the backward's non-GEMM work (ReLU masks, bias column sums, narrow heads: HIP kernels through
``ops``), at every batch size."""

from __future__ import annotations

import os
from collections.abc import Iterable, Sequence
from contextlib import contextmanager
from dataclasses import dataclass
from typing import Any

import torch
from torch import nn
from torch.nn.functional import linear

from cusrl_amd.utils.nest import iterate_nested

__all__ = ["Linear", "LinearFp32", "fused_inference_layers", "linear_act", "Mlp", "Module", "ModuleFactory", "disable_autocast",
           "double_differentiable", "is_unit_gradient", "register_unit_gradient", "resolve_activation_fn"]


def disable_autocast(device_type: str):
    return torch.autocast(device_type=device_type, enabled=False)


# While a sink is installed (ActorCritic._backward with a flat gradient buffer), split-batch weight gradients are
# handed over as their UNSUMMED [S, out, in] slabs, keyed by the weight's storage address, and the backward returns
# no weight gradient: the flat-buffer assembly sums the slabs straight into the parameter's slot (one launch for all
# parameters) instead of one sum(0) launch per layer followed by a concatenation.
_split_grad_sink: dict[int, torch.Tensor] | None = None


# The hand-written backward of ``_WideBatchLinear`` calls raw HIP kernels autograd cannot see, i.e. it is
# once-differentiable.  Code that differentiates THROUGH a backward (``torch.autograd.grad(..., create_graph=True)``:
# the AMP gradient penalty, cusrl/nn/layer/loss.py:10-56) wraps the forward in ``double_differentiable()`` so those
# layers take torch's own differentiable ops whatever the batch size; outside the context a second differentiation
# of the custom backward raises (``once_differentiable``) instead of silently treating it as a constant.
_plain_linear_depth = 0
# Rows from which a differentiated fp32 linear layer on the GPU takes ``_WideBatchLinear``: EVERY batch size since round 5.  Up
# to round 4 batches below 4096 rows went through torch's ``addmm`` backward, whose bias gradient is an ATen ``sum`` — a
# global reduce_kernel with a semaphore zeroed by a memset node once the batch has >= ~1024 rows, and a captured step
# containing those is not replayed reliably by this stack.  ``CUSRL_WIDE_LINEAR_MIN_ROWS`` restores
# a threshold for A/B runs and for the defect's reproduction (scripts/debug_amp_identity.py).
_WIDE_MIN_ROWS = int(os.environ.get("CUSRL_WIDE_LINEAR_MIN_ROWS", "1"))
# The first layer's backward as one launch (round 6, ops.input_layer_backward); CUSRL_INPUT_LAYER_KERNEL=0: the mask + column-sum
# pass followed by the split-batch weight-gradient GEMM of rounds 2-5 (A/B switch)
_INPUT_LAYER_KERNEL = os.environ.get("CUSRL_INPUT_LAYER_KERNEL", "1") != "0"


# Backward shortcuts for a UNIT incoming gradient.  The loss summands of a step are differentiated as separate roots with a
# persistent ones-scalar as their grad_output (ActorCritic._backward); a custom Function whose forward kernel already
# produced d loss / d input may hand that out unscaled — but only if the gradient that actually ARRIVES is that very
# scalar.  The engine passes a root's grad_output through untouched, so its address identifies it; anything a caller did
# to the loss (a hook re-weighting the objectives, GradScaler, accumulation) arrives as a different tensor and is
# multiplied in.  (Up to round 4 the shortcut was taken on a hint computed at forward time.)
# address -> weak reference of the registered scalar.  The entry lives exactly as long as the tensor does: while the tensor
# is alive nothing else can own its storage, so an equal address IS that tensor; when it dies (an agent dropped, a sweep
# building agent after agent) its address leaves the registry with it, and the allocator may hand the block to anything.
_unit_gradients: dict[int, "weakref.ref[torch.Tensor]"] = {}


def register_unit_gradient(ones: torch.Tensor) -> torch.Tensor:
    import weakref

    address = ones.data_ptr()
    _unit_gradients[address] = weakref.ref(ones)
    weakref.finalize(ones, _forget_unit_gradient, address)
    return ones


def _forget_unit_gradient(address: int) -> None:
    ref = _unit_gradients.get(address)
    if ref is not None and ref() is None:  # (a newer registration of a recycled address stays)
        del _unit_gradients[address]


def is_unit_gradient(grad: torch.Tensor | None) -> bool:
    if grad is None or grad.dim() != 0:
        return False
    ref = _unit_gradients.get(grad.data_ptr())
    return ref is not None and ref() is not None


@contextmanager
def double_differentiable():
    global _plain_linear_depth
    _plain_linear_depth += 1
    try:
        yield
    finally:
        _plain_linear_depth -= 1


@contextmanager
def collect_split_weight_grads():
    global _split_grad_sink
    previous, _split_grad_sink = _split_grad_sink, {}
    try:
        yield _split_grad_sink
    finally:
        _split_grad_sink = previous


def _hand_over(sink, key, gradient):
    """A gradient whose column sums are still pending (``ops.DeferredColumns``) goes to the flat-gradient assembly
    through the sink (and autograd gets no gradient for that parameter); without a sink, or when the parameter already
    left something there (used twice in the graph), it is reduced right here."""
    from cusrl_amd import ops

    if not isinstance(gradient, ops.DeferredColumns):
        return gradient
    if sink is not None and key is not None and key not in sink:
        sink[key] = gradient
        return None
    return gradient.materialize()


class _WideBatchLinear(torch.autograd.Function):
    """``linear(x, w, b)`` (optionally + ReLU in the GEMM epilogue) with a backward shaped for 256 CUs.

    * forward with ``relu``: ``torch._addmm_activation`` — bias and ReLU run in the hipBLASLt epilogue (the separate
      ReLU launch costs as much as the GEMM itself for the 48 -> 256 layer: 29.9 us vs 16.3 us measured);
    * dW = dY^T X has a tiny output (e.g. 128 x 256) and a huge reduction dim (the minibatch, 24 576): as ONE GEMM it
      yields a few dozen output tiles, i.e. most of the chip idles (rocBLAS picks no split-K here: 84 us measured).
      Splitting the batch into S slabs turns it into a batched GEMM with S x more workgroups plus a tiny sum
      (17 + 4 us) — still a rocBLAS/hipBLASLt MFMA GEMM, just with enough parallelism;
    * ReLU mask and bias gradient come from ONE HIP pass (``cusrl_relu_bwd_colsum``) instead of threshold_backward +
      a column-sum reduction (35 us -> ~15 us for [24576, 256]);
    * a head with <= 16 outputs (policy mean, value) gets dX, dW and db from ONE streaming pass
      (``cusrl_narrow_linear_bwd``) instead of two skinny GEMMs, a split-sum and a column sum — and when its input
      is the ReLU output of the layer in front, the same pass also performs that ReLU's backward mask and the
      bias-gradient column sums of that layer, so the layer's own epilogue launch is skipped.
    """

    @staticmethod
    def forward(ctx, input, weight, bias, splits, relu):
        if relu:
            output = torch._addmm_activation(bias, input, weight.t())
            ctx.save_for_backward(input, weight, output)
            output._cusrl_relu_output = True  # lets a narrow head behind it play this ReLU's backward (see backward)
        else:
            output = _one_output_linear(input, weight, bias)
            ctx.save_for_backward(input, weight)
        ctx.splits, ctx.relu, ctx.has_bias = splits, relu, bias is not None
        ctx.bias_key = bias.data_ptr() if bias is not None else None
        ctx.narrow = ((not relu) and input.is_contiguous() and weight.is_contiguous() and input.data_ptr() % 16 == 0
                      and _narrow_head(weight))
        ctx.input_is_relu_output = getattr(input, "_cusrl_relu_output", False)
        return output

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        from cusrl_amd import ops

        if ctx.relu:
            input, weight, output = ctx.saved_tensors
            sink = _split_grad_sink
            premasked = getattr(grad_output, "_cusrl_premasked", None)
            if (_INPUT_LAYER_KERNEL and premasked is None and not ctx.needs_input_grad[0] and ctx.needs_input_grad[1]
                    and ctx.has_bias and ctx.needs_input_grad[2]):
                grad_output = grad_output.contiguous()
                if ops.input_layer_supported(grad_output, output, input, weight):
                    # the bottom layer (its input — the observation — needs no gradient): ReLU mask, bias gradient and weight
                    # gradient from ONE pass over dY, Y and X, nothing written back per row (cusrl_input_layer_bwd)
                    d_weight, d_bias = ops.input_layer_backward(grad_output, output, input)
                    return None, d_weight, d_bias, None, None
            if premasked is not None and premasked[1] == grad_output._version and ctx.has_bias:
                grad_bias = premasked[0]  # the head behind this ReLU already masked its dX and summed its columns
            else:
                grad_output, grad_bias = ops.relu_backward_bias(
                    grad_output.contiguous(), output, defer=sink is not None and ctx.has_bias and ctx.needs_input_grad[2])
            grad_bias = _hand_over(sink, ctx.bias_key, grad_bias)
        else:
            input, weight = ctx.saved_tensors
            if ctx.narrow and ctx.needs_input_grad[1]:  # policy-mean / value head: dX, dW and db from one pass
                fuse_relu = ctx.input_is_relu_output and ctx.needs_input_grad[0]
                sink = _split_grad_sink
                grad_input, grad_weight, grad_bias, masked_colsum = ops.narrow_linear_backward(
                    grad_output.contiguous(), input, weight, need_input_grad=ctx.needs_input_grad[0], relu_input=fuse_relu,
                    defer=sink is not None)
                grad_weight = _hand_over(sink, weight.data_ptr(), grad_weight)
                grad_bias = _hand_over(sink, ctx.bias_key, grad_bias) if ctx.has_bias else None
                if fuse_relu:
                    # The input is a ReLU output, so masking dX by (input > 0) here IS that ReLU's backward (it is
                    # idempotent, so the producer may safely repeat it).  Tell the producer — through the tensor it
                    # will receive as grad_output — that the mask and the bias-gradient column sums are done; the
                    # version stamp voids the note if autograd accumulates another consumer's gradient in place.
                    grad_input._cusrl_premasked = (masked_colsum, grad_input._version)
                return grad_input, grad_weight, (grad_bias if ctx.has_bias else None), None, None
            grad_bias = None
            if ctx.has_bias and ctx.needs_input_grad[2]:
                _, grad_bias = ops.relu_backward_bias(grad_output.contiguous(), None)
        grad_input = grad_weight = None
        if ctx.needs_input_grad[0]:
            grad_input = grad_output @ weight
        if ctx.needs_input_grad[1]:
            rows, splits = input.shape[0], ctx.splits
            if splits > 1:
                gy = grad_output.reshape(splits, rows // splits, -1)
                slabs = torch.bmm(gy.transpose(1, 2), input.reshape(splits, rows // splits, -1))
                sink = _split_grad_sink
                if sink is not None and weight.data_ptr() not in sink:
                    sink[weight.data_ptr()] = slabs  # summed by the flat-gradient assembly
                else:
                    grad_weight = ops.sum_slabs(slabs)
            else:
                grad_weight = grad_output.t() @ input
        return grad_input, grad_weight, grad_bias, None, None


def _one_output_linear(input: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None) -> torch.Tensor:
    """``linear(input, weight, bias)``; a one-output head (the value head, a discriminator's logit) as ONE row-dot launch —
    torch's addmm has no bias epilogue for a one-column output and issues a broadcast-bias copy in front of a skinny GEMM."""
    from cusrl_amd import ops

    if weight.shape[0] == 1 and ops.narrow_linear_forward_supported(input, weight):
        return ops.narrow_linear_forward(input, weight, bias)
    return linear(input, weight, bias)


def _narrow_head(weight: torch.Tensor) -> bool:
    from cusrl_amd import ops

    return (weight.shape[0] <= 16 and weight.data_ptr() % 16 == 0
            and ops.narrow_linear_supported(weight.shape[1], weight.shape[0]))


def _batch_splits(rows: int) -> int:
    """Largest power-of-two slab count (<= 32) that divides the batch and leaves >= 1024 rows per slab."""
    if rows < 4096:
        return 1
    splits = 1
    while splits < 32 and rows % (splits * 2) == 0 and rows // (splits * 2) >= 1024:
        splits *= 2
    return splits


def _device_fp32(input: torch.Tensor, weight: torch.Tensor) -> bool:
    return (input.dim() == 2 and input.is_cuda and input.dtype == torch.float32 and weight.dtype == torch.float32
            and not torch.is_autocast_enabled("cuda"))


def linear_act(input: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None, relu: bool = False) -> torch.Tensor:
    """``relu?(linear(input, weight, bias))`` through the MI355X-shaped paths when the data is fp32 on the GPU."""
    if _device_fp32(input, weight) and (bias is not None or not relu):
        # every batch size (round 5: torch's addmm backward brings ATen's split `sum` — a memset node — into captured steps).
        # The custom backward is once-differentiable: code that differentiates through a backward (the AMP gradient penalty
        # of a non-ReLU discriminator) builds its forward inside `double_differentiable()` and gets torch's own ops here
        if (torch.is_grad_enabled() and (weight.requires_grad or input.requires_grad) and input.shape[0] >= _WIDE_MIN_ROWS
                and not _plain_linear_depth):
            return _WideBatchLinear.apply(input, weight, bias, _batch_splits(input.shape[0]), relu)
        if torch.is_grad_enabled() and (weight.requires_grad or input.requires_grad):
            output = linear(input, weight, bias)
            return torch.relu(output) if relu else output
        if relu:
            return torch._addmm_activation(bias, input, weight.t())
        if not torch.is_grad_enabled() or not (weight.requires_grad or input.requires_grad):
            return _one_output_linear(input, weight, bias)
    output = linear(input, weight, bias)
    return torch.relu(output) if relu else output


class Linear(nn.Linear):
    """``nn.Linear`` (same parameters, same state-dict keys) routed through :func:`linear_act`."""

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return linear_act(input, self.weight, self.bias)


class LinearFp32(nn.Linear):
    """Linear layer evaluated in fp32 even under autocast (layer/linear.py:12-16)."""

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        if not torch.is_autocast_enabled(input.device.type) and input.dtype == torch.float32:
            return Linear.forward(self, input)
        with disable_autocast(input.device.type):
            return linear(input.float(), self.weight.float(), None if self.bias is None else self.bias.float())


# The no-grad pass of backbone + head as ONE launch (round 6, ops.mlp2_forward; CUSRL_FUSED_INFERENCE=0: the library GEMM chain,
# A/B switch).  Acting on 4096 envs is three launch-bound GEMMs + the sampling launch: 24 of the 37 us a captured env step takes.
_FUSED_INFERENCE = os.environ.get("CUSRL_FUSED_INFERENCE", "1") != "0"


def fused_inference_layers(backbone, head: nn.Linear, input, memory=None):
    """``(w1, b1, w2, b2, w3, b3)`` when ``head(backbone(input))`` can be evaluated by ``ops.mlp2_forward`` right now: no gradient
    asked for, a feed-forward Linear / ReLU / Linear / ReLU backbone (``Mlp.inference_stack``) in front of a linear head, a
    contiguous fp32 ``[B, K]`` device input outside autocast, shapes the kernel takes; else None."""
    if (not _FUSED_INFERENCE or torch.is_grad_enabled() or memory is not None or not isinstance(backbone, Mlp)
            or not isinstance(head, nn.Linear) or not isinstance(input, torch.Tensor) or not input.is_cuda or input.dim() != 2
            or input.dtype != torch.float32 or torch.is_autocast_enabled("cuda")):
        return None
    stack = backbone.inference_stack()
    if stack is None:
        return None
    from cusrl_amd import ops

    layers = (*stack, head.weight, head.bias)
    return layers if ops.mlp2_forward_supported(input, layers) else None


class ModuleFactory:
    def __call__(self, input_dim: int | None = None, output_dim: int | None = None) -> "Module":
        raise NotImplementedError


class Module(nn.Module):
    """``nn.Module`` + declared dims, recurrent-memory helpers and an intermediate-representation dict
    (module/module.py:35-163)."""

    Factory = ModuleFactory

    def __init__(self, input_dim: int | None = None, output_dim: int | None = None, is_recurrent: bool = False,
                 like: "Module | None" = None, intermediate_repr: dict[str, Any] | None = None):
        super().__init__()
        if like is not None:
            input_dim, output_dim, is_recurrent = like.input_dim, like.output_dim, like.is_recurrent
        elif input_dim is None or output_dim is None:
            raise ValueError("'input_dim' and 'output_dim' must be specified when 'like' is not provided")
        elif input_dim <= 0:
            raise ValueError("'input_dim' must be a positive integer")
        elif output_dim <= 0:
            raise ValueError("'output_dim' must be a positive integer")
        self.input_dim, self.output_dim, self.is_recurrent = input_dim, output_dim, is_recurrent
        self.intermediate_repr: dict[str, Any] = intermediate_repr or {}
        self._rnn_compatible = False

    @property
    def device(self) -> torch.device:
        for tensor in itertools_chain(self.parameters(), self.buffers()):
            return tensor.device
        return torch.device("cpu")

    def step_memory(self, input, memory=None, **kwargs):
        if not self.is_recurrent:
            return None
        _, *next_memory = self(input, memory=memory, **kwargs)
        return next_memory[0]

    def reset_memory(self, memory, done=None):
        """Zero the recurrent state of finished envs in place."""
        if memory is None:
            return
        if isinstance(done, torch.Tensor):
            done = done.squeeze(-1)
        elif done is None:
            done = slice(None)
        for _, tensor in iterate_nested(memory):
            tensor[done] = 0

    def clear_intermediate_repr(self):
        self.intermediate_repr.clear()

    def rnn_compatible(self):
        """Let a feed-forward module be called like a recurrent one: ``module(x, memory=m) -> (y, m)``."""
        if not self.is_recurrent and not self._rnn_compatible:
            self._rnn_compatible = True
            plain_forward = self.forward

            def forward(input, **kwargs):
                output = plain_forward(input)
                return (output, kwargs["memory"]) if "memory" in kwargs else output

            self.forward = forward
        return self


def itertools_chain(*iterables):
    for it in iterables:
        yield from it


def resolve_activation_fn(activation_fn: str | type[nn.Module]) -> type[nn.Module]:
    if isinstance(activation_fn, str):
        name = activation_fn.removeprefix("torch.").removeprefix("nn.")
        resolved = getattr(nn, name, None)
        if resolved is None:
            raise ValueError(f"Unknown activation function '{activation_fn}'")
        return resolved
    return activation_fn


@dataclass(slots=True)
class MlpFactory(ModuleFactory):
    hidden_dims: Sequence[int]
    activation_fn: str | type[nn.Module] = "ReLU"
    ends_with_activation: bool = False
    dropout: float = 0.0

    def __call__(self, input_dim: int | None = None, output_dim: int | None = None):
        assert input_dim is not None
        return Mlp(input_dim, self.hidden_dims, output_dim, self.activation_fn, self.ends_with_activation, self.dropout)


class Mlp(Module):
    """Linear/activation stack; ``output_dim=None`` makes the last hidden width the output (module/mlp.py:31-93)."""

    Factory = MlpFactory

    def __init__(self, input_dim: int, hidden_dims: Iterable[int], output_dim: int | None = None,
                 activation_fn: str | type[nn.Module] = "ReLU", ends_with_activation: bool = False, dropout: float = 0.0):
        widths = list(hidden_dims) + ([] if output_dim is None else [output_dim])
        if not widths:
            raise ValueError("Mlp needs at least one layer")
        act = resolve_activation_fn(activation_fn)
        super().__init__(input_dim, widths[-1])
        layers: list[nn.Module] = []
        fan_in = input_dim
        for i, width in enumerate(widths):
            layers.append(Linear(fan_in, width))
            if i + 1 < len(widths) or ends_with_activation:
                layers.append(act())
                if dropout > 0.0:
                    layers.append(nn.Dropout(dropout))
            fan_in = width
        self.layers = nn.Sequential(*layers)

    def forward(self, input: torch.Tensor, **kwargs) -> torch.Tensor:
        layers = self.layers
        if not (input.is_cuda and input.dim() == 2):
            return layers(input)
        x, i, n = input, 0, len(layers)
        while i < n:  # Linear followed by ReLU runs as one GEMM with the ReLU in its epilogue
            module = layers[i]
            if type(module) is Linear and i + 1 < n and type(layers[i + 1]) is nn.ReLU:
                x = linear_act(x, module.weight, module.bias, relu=True)
                i += 2
            else:
                x = module(x)
                i += 1
        return x

    def __getitem__(self, index: int) -> nn.Module:
        return self.layers[index]

    def inference_stack(self):
        """``(w1, b1, w2, b2)`` when this is exactly Linear / ReLU / Linear / ReLU with biases (the preset's backbone,
        ``ends_with_activation=True``): the shape ``ops.mlp2_forward`` evaluates — together with the head behind it — in one
        launch when no gradient is asked for; None for anything else."""
        layers = self.layers
        if (len(layers) == 4 and type(layers[0]) is Linear and type(layers[1]) is nn.ReLU and type(layers[2]) is Linear
                and type(layers[3]) is nn.ReLU and layers[0].bias is not None and layers[2].bias is not None):
            return layers[0].weight, layers[0].bias, layers[2].weight, layers[2].bias
        return None
