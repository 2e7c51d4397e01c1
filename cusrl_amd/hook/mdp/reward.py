"""Affine reward shaping with optional clamping (counterpart of cusrl/hook/mdp/reward.py:10-47)."""

from __future__ import annotations

from cusrl_amd.template.hook import Hook

__all__ = ["RewardShaping"]


class RewardShaping(Hook):
    def __init__(self, scale: float = 1.0, shift: float = 0.0, lower_bound: float | None = None, upper_bound: float | None = None):
        super().__init__()
        self.scale, self.shift, self.lower_bound, self.upper_bound = scale, shift, lower_bound, upper_bound

    def post_step(self, transition):
        reward = transition["reward"]
        if self.scale != 1.0:
            reward.mul_(self.scale)
        if self.shift != 0.0:
            reward.add_(self.shift)
        if self.lower_bound is not None or self.upper_bound is not None:
            reward.clamp_(min=self.lower_bound, max=self.upper_bound)
