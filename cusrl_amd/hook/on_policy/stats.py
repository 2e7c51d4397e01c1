"""Post-update policy statistics (counterpart of cusrl/hook/on_policy/stats.py:10-40): KL between the behaviour
and the updated policy, importance-weighted advantage, action std — over batches from its own sampler."""

from __future__ import annotations

import torch

from cusrl_amd.template.buffer import Sampler
from cusrl_amd.template.hook import Hook

__all__ = ["OnPolicyStatistics"]


class OnPolicyStatistics(Hook):
    def __init__(self, sampler: Sampler | None = None):
        super().__init__(training_only=True)
        self.sampler = sampler if sampler is not None else Sampler()

    @torch.no_grad()
    def post_update(self):
        actor = self.agent.actor
        for _, batch in self.sampler(self.agent.buffer):
            with self.agent.autocast():
                action_dist, _ = actor(batch["observation"], memory=batch.get("actor_memory"), done=batch["done"])
            self.agent.record(kl_divergence=actor.compute_kl_div(batch["action_dist"], action_dist))
            logp_ratio = actor.compute_logp(action_dist, batch["action"]) - batch["action_logp"]
            self.agent.record(importance_weighted_advantage=batch["advantage"] * logp_ratio.exp())
            if "std" in action_dist:
                self.agent.record(action_std=action_dist["std"])
