"""Random Network Distillation bonus (counterpart of cusrl/hook/auxiliary/rnd.py:15-81): a frozen random target net
and a trained predictor; the prediction error on ``next_state`` is added to the buffer's reward before the value
target / GAE hooks run, and minimised as ``rnd_loss``.  The two MLPs stay torch modules; the reward epilogue is one HIP
launch (``cusrl_rnd_reward``)."""

from __future__ import annotations

import itertools

import torch
from torch import nn

from cusrl_amd.template.hook import Hook
from cusrl_amd.utils.misc import get_first, host_form

__all__ = ["RandomNetworkDistillation"]


class _MseLossFunction(torch.autograd.Function):
    """``nn.MSELoss()(prediction, target)`` with the target constant: loss AND d loss / d prediction from one pass
    (``cusrl_mse_loss_fwd_bwd``) instead of sub / square / mean forward and fill / mse_backward behind it."""

    @staticmethod
    def forward(ctx, prediction, target):
        from cusrl_amd import ops

        loss, grad = ops.mse_loss_fwd_bwd(prediction, target)
        ctx.save_for_backward(grad)
        ctx.shape = prediction.shape
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_loss):
        from cusrl_amd.nn.module import is_unit_gradient

        (grad,) = ctx.saved_tensors
        if not is_unit_gradient(grad_loss):  # GradScaler, or a caller that rescales the loss: anything but the agent's unit scalar
            grad = grad * grad_loss
        return grad.view(ctx.shape), None


class RandomNetworkDistillation(Hook):
    def __init__(self, module_factory, output_dim: int, reward_scale: float, state_indices=None):
        super().__init__()
        self.output_dim = output_dim
        self.module_factory = module_factory
        self.state_indices = slice(None) if state_indices is None else state_indices
        self.reward_scale: float = reward_scale
        self.register_mutable("reward_scale")

    def init(self):
        input_dim = torch.ones(1, self.agent.state_dim)[..., self.state_indices].numel()
        target, predictor = self.module_factory(input_dim, self.output_dim), self.module_factory(input_dim, self.output_dim)
        for module in itertools.chain(target.modules(), predictor.modules()):
            if isinstance(module, nn.Linear):
                nn.init.xavier_normal_(module.weight)
                nn.init.zeros_(module.bias)
        self.register_module("target", target)
        self.register_module("predictor", predictor)
        self.target.requires_grad_(False)
        self.criterion = nn.MSELoss()

    @torch.no_grad()
    def pre_update(self, buffer):
        next_state = get_first(buffer, "next_state", "next_observation")[..., self.state_indices]
        flat = next_state.reshape(-1, next_state.shape[-1])
        target, prediction = self.target(flat), self.predictor(flat)
        reward = buffer["reward"]
        if reward.is_cuda:
            # on the GPU the epilogue is ALWAYS the HIP kernel: a layout it cannot take is brought into shape (one
            # reward channel at a time, contiguous staging) rather than silently computed by torch ops
            from cusrl_amd import ops

            if reward.shape[-1] == 1 and reward.is_contiguous():
                bonus = ops.rnd_reward_(reward, target, prediction, self.reward_scale)
            else:  # multi-channel / strided rewards: the same bonus is added to every channel (broadcast, rnd.py:71-74)
                staged = reward.new_zeros(reward.shape[:-1] + (1,)).contiguous()
                bonus = ops.rnd_reward_(staged, target, prediction, self.reward_scale)
                reward.add_(bonus)
        else:  # test processes without a GPU only: the reference's torch ops
            host_form("RandomNetworkDistillation.pre_update")
            bonus = self.reward_scale * (target - prediction).square().mean(dim=-1, keepdim=True).view(*reward.shape[:-1], 1)
            reward.add_(bonus)
        self.agent.record(rnd_reward=bonus)

    def objective(self, metadata, batch):
        next_state = get_first(batch, "next_state", "next_observation")[..., self.state_indices]
        prediction, target = self.predictor(next_state), self.target(next_state)
        criterion = self.criterion
        if (prediction.is_cuda and type(criterion) is nn.MSELoss and criterion.reduction == "mean"
                and prediction.dtype == torch.float32 and target.dtype == torch.float32 and not target.requires_grad):
            # forward + backward of the squared error in one HIP pass; a user-supplied criterion keeps torch's ops
            return {"rnd_loss": _MseLossFunction.apply(prediction, target)}
        return {"rnd_loss": criterion(prediction, target)}
