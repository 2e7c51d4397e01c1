"""Policy re-evaluation on a minibatch (replaces cusrl/hook/on_policy/common.py:12-49): actor forward, then
log-prob, entropy and the probability ratio against the behaviour policy."""

from __future__ import annotations

from cusrl_amd.hook.on_policy.fused import FusedPpoObjective
from cusrl_amd.template.hook import Hook

__all__ = ["OnPolicyPreparation"]


class OnPolicyPreparation(Hook):
    def __init__(self, calculate_kl_divergence: bool = False):
        super().__init__(training_only=True)
        self.calculate_kl_divergence = calculate_kl_divergence

    def objective(self, metadata, batch):
        actor = self.agent.actor
        action_dist, _ = actor(batch["observation"], memory=batch.get("actor_memory"), done=batch["done"])
        batch["curr_action_dist"] = action_dist
        if self.calculate_kl_divergence:
            batch["kl_divergence"] = actor.compute_kl_div(batch["action_dist"], action_dist)
        if (fused := FusedPpoObjective.current(self)) is not None:
            # logp / entropy / ratios (and their gradients) come out of the fused kernel at resolve time
            # (split mode — further objective hooks present: add_policy also leaves them in the batch right now, as
            # differentiable tensors from one cusrl_policy_terms_fwd launch)
            fused.add_policy(action_dist, batch["action"], batch["action_logp"], batch)
            return None
        action_logp = actor.compute_logp(action_dist, batch["action"])
        logp_ratio = action_logp - batch["action_logp"]
        batch["curr_action_logp"] = action_logp
        batch["curr_entropy"] = actor.compute_entropy(action_dist)
        batch["action_logp_ratio"] = logp_ratio
        batch["action_prob_ratio"] = logp_ratio.exp()
        return None

    def post_objective(self, metadata, batch):
        if (reduced := batch.get("_fused_metrics")) is not None:
            if reduced.get("deferred"):  # captured step: the kernel's running sums are read once per update (ops.DeferredLoss)
                return
            self.agent.metrics.record_reduced("ratio", *reduced["ratio"])
            self.agent.metrics.record_reduced("entropy", *reduced["entropy"])
        else:
            self.agent.record(ratio=batch["action_logp_ratio"].abs(), entropy=batch["curr_entropy"])
