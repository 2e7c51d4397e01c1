"""Adversarial Motion Priors (counterpart of cusrl/hook/auxiliary/amp.py:16-168): a discriminator separates agent
transitions from expert transitions; its verdict becomes a style reward added to every env step, and the buffer carries
the normalised ``agent_transition`` / ``expert_transition`` leaves the discriminator is trained on.  Discriminator and
running statistics stay torch / RunningMeanStd; the style-reward epilogue is one HIP launch
(``cusrl_amp_style_reward``)."""

from __future__ import annotations

import os

import numpy as np
import torch
from torch import Tensor, nn

from cusrl_amd.nn.rms import RunningMeanStd
from cusrl_amd.template.hook import Hook
from cusrl_amd.utils.misc import get_first, host_form

__all__ = ["AdversarialMotionPrior", "GradientPenaltyLoss"]


class GradientPenaltyLoss(nn.Module):
    """``E[ || d outputs / d inputs ||^2 ]`` (cusrl/nn/layer/loss.py:10-56); ``inputs`` must require grad."""

    def __init__(self, reduction: str = "mean"):
        super().__init__()
        self.reduction = reduction

    def forward(self, outputs: Tensor, inputs: Tensor) -> Tensor:
        (gradients,) = torch.autograd.grad(outputs, inputs, grad_outputs=torch.ones_like(outputs), create_graph=True,
                                           retain_graph=True, only_inputs=True)
        penalty = gradients.square().sum(dim=-1)
        if self.reduction == "mean":
            return penalty.mean()
        return penalty.sum() if self.reduction == "sum" else penalty


def _hidden(x: Tensor, weight: Tensor, bias: Tensor) -> Tensor:
    if x.is_cuda:
        return torch._addmm_activation(bias, x, weight.t())  # the ReLU in the GEMM's epilogue
    return torch.relu(torch.addmm(bias, x, weight.t()))


_threshold = torch.ops.aten.threshold_backward  # (gradient, activation, 0): the gradient where the ReLU let the value through


def _device_rows(gradient: Tensor, activation: Tensor) -> bool:
    return (gradient.is_cuda and gradient.dtype == torch.float32 and activation.dtype == torch.float32 and gradient.dim() == 2
            and gradient.is_contiguous() and activation.is_contiguous() and gradient.shape == activation.shape
            and gradient.data_ptr() % 16 == 0 and activation.data_ptr() % 16 == 0)


def _masked(gradient: Tensor, activation: Tensor, _threshold_value: int = 0) -> Tensor:
    """``gradient`` where ``activation > 0`` else 0 — on the device through the MLP backward's one-pass mask kernel
    (``cusrl_relu_bwd_colsum``; its column sums ride along unused), else torch's ``threshold_backward``."""
    if _device_rows(gradient, activation):
        from cusrl_amd import ops

        return ops.relu_backward_bias(gradient, activation, defer=True)[0]
    return _threshold(gradient, activation, 0)


def _masked_with_bias(gradient: Tensor, activation: Tensor, bias_key: int, ones: Tensor):
    """``(masked gradient, its column sums = the gradient of the bias in front of that ReLU)``.  On the device both come from
    ONE launch, and with the agent's flat gradient buffer open (``nn/module.py::_split_grad_sink``) the column sums stay the
    kernel's partial rows, handed to the gradient assembly under the bias parameter's address (then ``None`` is returned for
    them) — instead of a mask launch plus a ``ones @ d_out`` GEMM."""
    if _device_rows(gradient, activation):
        from cusrl_amd import ops
        from cusrl_amd.nn import module as nn_module

        sink = nn_module._split_grad_sink
        masked, colsum = ops.relu_backward_bias(gradient, activation, defer=sink is not None)
        return masked, nn_module._hand_over(sink, bias_key, colsum)
    masked = _threshold(gradient, activation, 0)
    return masked, (ones @ masked).reshape(-1)


class _ReluDiscriminatorObjective(torch.autograd.Function):
    """Both AMP terms of a Linear / ReLU discriminator in closed form (amp.py:135-154 with loss.py:10-56 behind it):

    ``discrimination = (BCE(D(agent), 0) + BCE(D(expert), 1)) / 2`` and ``penalty = mean_n || dD(expert_n)/d expert_n ||^2``.

    With ReLU units the input gradient is ``W_1^T (m_1 * (W_2^T (m_2 * ... W_last^T)))`` (``m_k`` = the units that
    fired), linear in every weight matrix, and ``relu'' = 0``: autograd's double backward through the discriminator
    evaluates exactly these products, one small kernel at a time, plus the bookkeeping of three separate graphs (agent
    pass, expert pass, penalty) whose weight gradients it then adds up.  Here the two passes share one ``[2N, .]`` batch
    and every weight gradient is produced once: ~40 launches instead of ~80 per minibatch step, the same arithmetic."""

    @staticmethod
    def forward(ctx, agent, expert, target, ones, loss_weight, penalty_weight, *parameters):
        """``expert is None``: ``agent`` already is the joint ``[2N, C]`` batch (agent rows first), e.g. filled by ONE row
        gather of both buffer leaves."""
        weights, biases = parameters[0::2], parameters[1::2]
        rows = agent.shape[0] // 2 if expert is None else expert.shape[0]
        hidden = [agent if expert is None else torch.cat((agent, expert))]
        for weight, bias in zip(weights[:-1], biases[:-1]):
            hidden.append(_hidden(hidden[-1], weight, bias))
        if hidden[-1].is_cuda and weights[-1].shape[0] == 1:
            from cusrl_amd.nn.module import _one_output_linear

            logit = _one_output_linear(hidden[-1], weights[-1], biases[-1])  # one row-dot launch (no broadcast-bias copy + GEMM)
        else:
            logit = torch.addmm(biases[-1], hidden[-1], weights[-1].t())
        on_device = logit.is_cuda and logit.dtype == torch.float32
        if on_device:
            # loss AND d loss / d logit of the joint batch from ONE launch (log_sigmoid / mul / add / mean forward and
            # sigmoid / sub / mul backward as torch ops: ~10 launches over 1024 logits)
            from cusrl_amd import ops

            discrimination, d_logit = ops.bce_pair_fwd_bwd(logit, loss_weight)
        else:
            discrimination, d_logit = nn.functional.binary_cross_entropy_with_logits(logit, target) * loss_weight, None
        # d logit / d expert, layer by layer from the output: u_k = m_k * (u_{k+1} W_{k+1})
        units = [_threshold(weights[-1].expand(rows, -1), hidden[-1][rows:], 0)]  # (a broadcast operand: torch's kernel)
        for weight, activation in zip(reversed(weights[1:-1]), reversed(hidden[1:-1])):
            units.append(_masked(units[-1] @ weight, activation[rows:], 0))
        units.reverse()  # units[k - 1] = u_k
        input_gradient = units[0] @ weights[0]
        if on_device:
            # penalty = mean_n ||g_n||^2 and what it sends back, 2 g / rows — both with their weights — from one pass
            penalty, d_input = ops.sumsq_fwd_bwd(input_gradient, penalty_weight * loss_weight / rows,
                                                 2.0 * penalty_weight * loss_weight / rows)
            ctx.save_for_backward(d_logit, target, ones, d_input, *hidden, *units, *weights)
        else:
            flat = input_gradient.reshape(-1)
            penalty = torch.dot(flat, flat) / rows * (penalty_weight * loss_weight)
            ctx.save_for_backward(logit, target, ones, input_gradient, *hidden, *units, *weights)
        ctx.on_device = on_device
        ctx.layers, ctx.rows = len(weights), rows
        ctx.bias_keys = tuple(bias.data_ptr() for bias in biases)  # the backward hands deferred bias column sums over under them
        ctx.loss_weight, ctx.penalty_weight = loss_weight, penalty_weight
        return discrimination, penalty

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_discrimination, grad_penalty):
        layers, rows = ctx.layers, ctx.rows
        logit, target, ones, input_gradient, *saved = ctx.saved_tensors  # ones: [1, 2N], column sums as GEMMs
        hidden, units, weights = saved[:layers], saved[layers:2 * layers - 1], saved[2 * layers - 1:]
        # --- penalty: d/dW_k of mean || u_1 W_1 ||^2, the masks being constants
        from cusrl_amd.nn.module import is_unit_gradient

        if ctx.on_device:  # `input_gradient` already is 2 g pw lw / rows, `logit` already d loss / d logit (forward kernels):
            # used as they are when the incoming gradient IS the agent's unit scalar (the summands as separate roots), else scaled
            d_input = input_gradient if is_unit_gradient(grad_penalty) else input_gradient * grad_penalty
        else:
            d_input = input_gradient * (grad_penalty * (2.0 * ctx.penalty_weight * ctx.loss_weight / rows))
        penalty_grads = [units[0].t() @ d_input]
        d_units = d_input @ weights[0].t()
        for k in range(1, layers - 1):
            d_pre = _masked(d_units, hidden[k][rows:], 0)  # through m_k, to v_k = u_{k+1} W_{k+1}
            penalty_grads.append(units[k].t() @ d_pre)
            d_units = d_pre @ weights[k].t()
        penalty_grads.append(ones[:, :rows] @ _masked(d_units, hidden[layers - 1][rows:], 0))
        # --- discrimination: an ordinary MLP backward over the joint batch, the penalty's share added by the GEMM
        if ctx.on_device:
            d_out = logit if is_unit_gradient(grad_discrimination) else logit * grad_discrimination
        else:
            d_out = (torch.sigmoid(logit) - target) * (grad_discrimination * (ctx.loss_weight / logit.shape[0]))
        gradients: list[Tensor | None] = []
        bias_gradient: Tensor | None = (ones @ d_out).reshape(-1)  # the logit's bias: the column sum of a [2N, 1] matrix
        for k in range(layers - 1, -1, -1):
            gradients.append(bias_gradient)
            gradients.append(torch.addmm(penalty_grads[k], d_out.t(), hidden[k]))
            if k:  # through the ReLU in front of layer k: mask AND the bias gradient of layer k - 1 from one launch
                d_out, bias_gradient = _masked_with_bias(d_out @ weights[k], hidden[k], ctx.bias_keys[k - 1], ones)
        gradients.reverse()  # weight_1, bias_1, weight_2, ...
        return (None, None, None, None, None, None, *gradients)


class AdversarialMotionPrior(Hook):
    objective_draws_random = True  # torch.randint for the discriminator batch
    step_draws_random = True  # torch.randint for the step's expert transitions (post_step; the reference's amp.py:161)
    # Extension: a Linear / ReLU discriminator takes the closed-form objective above (CUSRL_AMP_CLOSED_FORM=0 or this
    # attribute restore the autograd double backward, which any other discriminator keeps anyway).
    closed_form_objective: bool = os.environ.get("CUSRL_AMP_CLOSED_FORM", "1") != "0"

    def __init__(self, discriminator_factory, dataset_source=None, state_indices=None, batch_size: int | None = 512,
                 reward_scale: float = 1.0, loss_weight: float = 1.0, grad_penalty_weight: float = 5.0):
        super().__init__()
        self.discriminator_factory = discriminator_factory
        self.dataset_source = dataset_source
        self.state_indices = state_indices
        self.batch_size, self.reward_scale = batch_size, reward_scale
        self.loss_weight, self.grad_penalty_weight = loss_weight, grad_penalty_weight
        for name in ("batch_size", "reward_scale", "loss_weight", "grad_penalty_weight"):
            self.register_mutable(name)
        self.dataset: Tensor | None = None
        self._targets: Tensor | None = None
        self._ones: Tensor | None = None
        self._columns: tuple | None = None  # (state width, prefix width | None, int32 device column vector | None)

    def init(self):
        source = self.dataset_source
        if isinstance(source, str):
            if source.endswith(".npy"):
                self.dataset = torch.as_tensor(np.load(source), device=self.agent.device)
            elif source.endswith(".pt"):
                self.dataset = torch.load(source, map_location=self.agent.device)
            else:
                raise ValueError(f"Unsupported dataset file format for '{source}'")
        elif isinstance(source, (Tensor, np.ndarray)):
            self.dataset = self.agent.to_tensor(source)
        elif callable(source):
            self.dataset = self.agent.to_tensor(source())
        elif source is not None:
            raise ValueError(f"Unsupported 'dataset_source' type: {type(source)}")
        self.transition_dim = self._sample_demonstration(1).size(-1)
        self.register_module("discriminator", self.discriminator_factory(self.transition_dim, 1))
        self.register_module("transition_rms", RunningMeanStd(self.transition_dim))
        self.criterion = nn.BCEWithLogitsLoss()
        self.grad_penalty = GradientPenaltyLoss()

    def collective_phases(self):
        return ("step",)  # transition_rms.update synchronises across ranks on every env step (amp.py:123-124)

    def _selected_columns(self, width: int, device):
        """``state_indices`` as what the one-launch preparation takes: a prefix width, or an int32 device column vector."""
        cached = self._columns
        if cached is None or cached[0] != width:
            picked = torch.arange(width)[self.state_indices].reshape(-1)
            prefix = picked.numel() if torch.equal(picked, torch.arange(picked.numel())) else None
            cached = self._columns = (width, prefix, None if prefix is not None else picked.to(device=device, dtype=torch.int32))
        return cached[1], cached[2]

    def _prepare_fused(self, transition, agent_raw):
        """``state[idx] || next_state[idx]`` + ``dataset[randint]`` + both statistics updates + both normalisations as ONE
        launch (``cusrl_amp_prepare``), or None when a condition of the fused form does not hold (grouped / excluded
        statistics channels, several ranks synchronising every update, layouts it does not take) — the caller then issues
        the same steps as separate HIP launches."""
        from cusrl_amd import ops
        from cusrl_amd.utils import distributed

        rms = self.transition_rms
        if not rms.mean.is_cuda or rms.groups or rms.excluded_indices is not None or distributed.enabled():
            return None
        kwargs = {}
        if agent_raw is not None:
            if not (agent_raw.is_cuda and agent_raw.dim() == 2 and agent_raw.dtype == torch.float32):
                return None
            rows, channels = agent_raw.shape
            kwargs["agent_raw"] = agent_raw
        else:
            state = get_first(transition, "state", "observation")
            next_state = get_first(transition, "next_state", "next_observation")
            if not (state.is_cuda and state.dim() == 2 and state.dtype == torch.float32 and state.shape == next_state.shape
                    and next_state.dtype == torch.float32):
                return None
            prefix, columns = self._selected_columns(state.shape[1], state.device)
            rows, channels = state.shape[0], 2 * (prefix if prefix is not None else columns.numel())
            kwargs.update(state=state, next_state=next_state, columns=columns, width=prefix)
        if channels != self.transition_dim or not ops.amp_prepare_supported(rows, channels):
            return None
        stock_sampler = ("_sample_demonstration" not in self.__dict__
                         and type(self)._sample_demonstration is AdversarialMotionPrior._sample_demonstration)
        if self.dataset is not None and stock_sampler:
            if not (self.dataset.is_cuda and self.dataset.dtype == torch.float32 and self.dataset.dim() == 2):
                return None
            # the reference's draw (amp.py:161), so the random stream stays the reference's
            kwargs.update(dataset=self.dataset, indices=torch.randint(self.dataset.size(0), (rows,), device=self.agent.device))
        else:
            kwargs["expert_raw"] = self._sample_demonstration(rows).float()
        return ops.amp_prepare(rms, **kwargs)

    @torch.no_grad()
    def post_step(self, transition):
        agent_transition = transition.pop("amp_obs", None)
        if agent_transition is None and self.state_indices is None:
            raise ValueError("AMP observations were not provided, and 'state_indices' is not set")
        prepared = self._prepare_fused(transition, agent_transition) if transition["reward"].is_cuda else None
        if prepared is not None:
            agent_transition, expert_transition = prepared
            self.transition_rms._is_synchronized = True
        else:
            if agent_transition is None:
                state = get_first(transition, "state", "observation")[..., self.state_indices]
                next_state = get_first(transition, "next_state", "next_observation")[..., self.state_indices]
                agent_transition = torch.cat([state, next_state], dim=-1)
            expert_transition = self._sample_demonstration(agent_transition.size(0))
            self.transition_rms.update(agent_transition)
            self.transition_rms.update(expert_transition)
            agent_transition = self.transition_rms.normalize(agent_transition)
            expert_transition = self.transition_rms.normalize(expert_transition)
        transition["agent_transition"] = agent_transition
        transition["expert_transition"] = expert_transition
        logit = self.discriminator(agent_transition)
        reward = transition["reward"]
        if reward.is_cuda:
            # on the GPU the style-reward epilogue is ALWAYS the HIP kernel; a reward it cannot update in place (strided,
            # several channels) gets the kernel's bonus added instead of a silent torch-op evaluation
            from cusrl_amd import ops

            if reward.is_contiguous() and reward.shape == logit.shape and reward.dtype == torch.float32:
                # bonus, reward update and the mean the metric records: one launch
                metrics = getattr(self.agent, "metrics", None)
                if hasattr(metrics, "record_reduced"):
                    style_reward, mean = ops.amp_style_reward_mean_(reward, logit, self.reward_scale)
                    metrics.record_reduced("amp_reward", mean[0], style_reward.numel())
                    return
                style_reward = ops.amp_style_reward_(reward, logit, self.reward_scale)
            else:
                style_reward = ops.amp_style_reward_(torch.zeros_like(logit, dtype=torch.float32), logit.float(), self.reward_scale)
                reward.add_(style_reward.to(reward.dtype))
        else:  # test processes without a GPU only: the reference's torch ops
            host_form("AdversarialMotionPrior.post_step")
            style_reward = self.reward_scale * -torch.log(torch.clamp(1 - 1 / (1 + torch.exp(-logit)), min=1e-4))
            reward.add_(style_reward)
        self.agent.record(amp_reward=style_reward)

    def objective(self, metadata, batch):
        agent_transition = batch["agent_transition"].flatten(0, -2)
        expert_transition = batch["expert_transition"].flatten(0, -2)
        parameters = self._relu_stack() if self.closed_form_objective else None
        closed_form = parameters is not None and all(p.dtype == agent_transition.dtype for p in parameters)
        joint = None
        if self.batch_size is not None:
            indices = torch.randint(agent_transition.size(0), (self.batch_size,), device=self.agent.device)
            if (agent_transition.is_cuda and agent_transition.is_contiguous() and expert_transition.is_contiguous()
                    and agent_transition.dtype == torch.float32 and expert_transition.dtype == torch.float32):
                # both subsamples with ONE launch of the row-gather kernel — and, for the closed-form objective, straight into
                # the joint [2N, C] batch it evaluates (no index kernels, no cat)
                from cusrl_amd import ops

                pool, rows, width = agent_transition.size(0), self.batch_size, agent_transition.size(-1)
                joint = torch.empty((2 * rows, width), dtype=torch.float32, device=agent_transition.device)
                ops.gather_rows([agent_transition.view(1, pool, width), expert_transition.view(1, pool, width)], indices, 1, pool,
                                out=[joint[:rows], joint[rows:]])
                agent_transition, expert_transition = joint[:rows], joint[rows:]
            else:
                agent_transition, expert_transition = agent_transition[indices], expert_transition[indices]
        if closed_form:
            rows = agent_transition.size(0)
            if (self._targets is None or self._targets.size(0) != 2 * rows or self._targets.device != agent_transition.device
                    or self._targets.dtype != agent_transition.dtype):
                self._targets = torch.cat((agent_transition.new_zeros(rows, 1), agent_transition.new_ones(rows, 1)))
                self._ones = agent_transition.new_ones(1, 2 * rows)
            discrimination, penalty = _ReluDiscriminatorObjective.apply(
                agent_transition if joint is None else joint, expert_transition if joint is None else None, self._targets,
                self._ones, self.loss_weight, self.grad_penalty_weight, *parameters)
            return {"amp_discrimination_loss": discrimination, "amp_grad_penalty_loss": penalty}
        expert_transition.requires_grad_(True)
        from cusrl_amd.nn.module import double_differentiable

        agent_logit = self.discriminator(agent_transition)
        with double_differentiable():  # the gradient penalty differentiates through this forward's backward
            expert_logit = self.discriminator(expert_transition)
        discrimination = (self.criterion(agent_logit, torch.zeros_like(agent_logit))
                          + self.criterion(expert_logit, torch.ones_like(expert_logit))) / 2
        penalty = self.grad_penalty(expert_logit, expert_transition)
        return {
            "amp_discrimination_loss": discrimination * self.loss_weight,
            "amp_grad_penalty_loss": penalty * (self.grad_penalty_weight * self.loss_weight),
        }

    def _relu_stack(self) -> list[Tensor] | None:
        """``[weight_1, bias_1, weight_2, ...]`` when the discriminator is Linear (ReLU Linear)* -> 1 logit, the
        terms are the stock ones, and nothing (autocast, a dropout layer, a missing bias) changes what a pass computes."""
        from cusrl_amd.nn.module import Mlp

        module = self.discriminator
        if (not isinstance(module, Mlp) or type(self.criterion) is not nn.BCEWithLogitsLoss
                or self.criterion.weight is not None or self.criterion.pos_weight is not None
                or self.criterion.reduction != "mean" or type(self.grad_penalty) is not GradientPenaltyLoss
                or self.grad_penalty.reduction != "mean" or torch.is_autocast_enabled(self.agent.device.type)):
            return None
        layers = list(module.layers)
        if len(layers) < 3 or len(layers) % 2 == 0:
            return None
        parameters: list[Tensor] = []
        for index, layer in enumerate(layers):
            if index % 2:
                if type(layer) is not nn.ReLU:
                    return None
            elif not isinstance(layer, nn.Linear) or layer.bias is None:
                return None
            else:
                parameters += [layer.weight, layer.bias]
        return parameters if layers[-1].out_features == 1 else None

    def _sample_demonstration(self, num_samples: int) -> Tensor:
        if self.dataset is not None:
            indices = torch.randint(self.dataset.size(0), (num_samples,), device=self.agent.device)
            return self.dataset[indices]
        sampler = self.agent.environment_spec.demonstration_sampler
        if sampler is None:
            raise ValueError("Provide either 'dataset_source' or 'environment_spec.demonstration_sampler'")
        return self.agent.to_tensor(sampler(num_samples))
