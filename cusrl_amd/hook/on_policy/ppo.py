"""Clipped-surrogate and entropy terms (replaces cusrl/hook/on_policy/ppo.py:10-84).

Inside the stock PPO composition both hooks only register their term with the armed
:class:`~cusrl_amd.hook.on_policy.fused.FusedPpoObjective` (ONE HIP launch computes every loss and gradient);
standalone they evaluate the same formulas hook by hook.
"""

from __future__ import annotations

import torch

from cusrl_amd.hook.on_policy.fused import FusedPpoObjective
from cusrl_amd.template.hook import Hook

__all__ = ["EntropyLoss", "PpoSurrogateLoss"]


def _ppo_surrogate_loss(advantage: torch.Tensor, prob_ratio: torch.Tensor, clip_ratio: float) -> torch.Tensor:
    """``-mean(min(A r, A clamp(r, 1 - eps, 1 + eps)))`` (ppo.py:10-18)."""
    clipped = prob_ratio.clamp(1.0 - clip_ratio, 1.0 + clip_ratio)
    return -torch.min(advantage * prob_ratio, advantage * clipped).mean()


class PpoSurrogateLoss(Hook):
    def __init__(self, clip_ratio: float = 0.2, weight: float = 1.0):
        if clip_ratio <= 0:
            raise ValueError("'clip_ratio' must be positive")
        if weight < 0:
            raise ValueError("'weight' must be non-negative")
        super().__init__(training_only=True)
        self.clip_ratio: float = clip_ratio
        self.weight: float = weight
        self.register_mutable("clip_ratio")
        self.register_mutable("weight")

    def objective(self, metadata, batch):
        advantage = batch["advantage"]
        if advantage.size(-1) != 1:
            raise ValueError(f"Expected advantage to have shape [..., 1], got {advantage.shape}")
        if (fused := FusedPpoObjective.current(self)) is not None:
            if fused.owns(batch, "action_prob_ratio"):
                return fused.add_surrogate(advantage, self.clip_ratio, self.weight)
            fused.drop_surrogate()  # a further hook replaced the ratio: this term is evaluated from what it left
        loss = _ppo_surrogate_loss(advantage, batch["action_prob_ratio"], self.clip_ratio)
        return {"surrogate_loss": loss * self.weight}


class EntropyLoss(Hook):
    def __init__(self, weight: float = 0.01):
        if weight < 0:
            raise ValueError("'weight' must be non-negative")
        super().__init__(training_only=True)
        self.weight: float = weight
        self.register_mutable("weight")

    def objective(self, metadata, batch):
        if (fused := FusedPpoObjective.current(self)) is not None:
            if fused.owns(batch, "curr_entropy"):
                return fused.add_entropy(self.weight)
            fused.drop_entropy()
        return {"entropy_loss": -batch["curr_entropy"].mean() * self.weight}
