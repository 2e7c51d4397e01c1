"""Affine reward shaping with optional clamping (counterpart of cusrl/hook/mdp/reward.py:10-47): on a GPU one HIP launch
(``cusrl_reward_shaping``) for the reference's ``mul_`` / ``add_`` / ``clamp_`` chain."""

from __future__ import annotations

import torch

from cusrl_amd.template.hook import Hook
from cusrl_amd.utils.misc import host_form

__all__ = ["RewardShaping"]


class RewardShaping(Hook):
    def __init__(self, scale: float = 1.0, shift: float = 0.0, lower_bound: float | None = None, upper_bound: float | None = None):
        super().__init__()
        self.scale, self.shift, self.lower_bound, self.upper_bound = scale, shift, lower_bound, upper_bound

    def post_step(self, transition):
        reward = transition["reward"]
        bounded = self.lower_bound is not None or self.upper_bound is not None
        if self.scale == 1.0 and self.shift == 0.0 and not bounded:
            return  # the identity: nothing to launch
        if isinstance(reward, torch.Tensor) and reward.is_cuda:
            from cusrl_amd import ops

            if reward.dtype == torch.float32 and reward.is_contiguous():
                ops.reward_shaping_(reward, self.scale, self.shift, self.lower_bound, self.upper_bound)
            else:  # a layout the launch does not take is staged, not handed to torch's elementwise ops
                staged = reward.float().contiguous()
                ops.reward_shaping_(staged, self.scale, self.shift, self.lower_bound, self.upper_bound)
                reward.copy_(staged)
            return
        host_form("RewardShaping.post_step")  # test processes without a GPU only
        reward.mul_(self.scale).add_(self.shift)
        if bounded:
            reward.clamp_(min=self.lower_bound, max=self.upper_bound)
