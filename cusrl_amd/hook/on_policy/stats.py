"""Post-update policy statistics (counterpart of cusrl/hook/on_policy/stats.py:10-40).

After every update the refreshed actor is evaluated on the batches of the hook's own sampler and three means are
recorded: ``kl_divergence`` (behaviour policy stored in the buffer vs the updated policy),
``importance_weighted_advantage`` and ``action_std``.  For a Gaussian policy on the GPU the three reductions come
from ONE pass over the batch (``cusrl_policy_stats``: the actor's GEMMs are the only other device work of this
hook); any other distribution family takes the op-by-op form through the actor's own ``compute_*`` methods.
"""

from __future__ import annotations

import torch

from cusrl_amd import ops
from cusrl_amd.template.buffer import Sampler
from cusrl_amd.template.hook import Hook

__all__ = ["OnPolicyStatistics"]


class OnPolicyStatistics(Hook):
    def __init__(self, sampler: Sampler | None = None):
        super().__init__(training_only=True)
        self.sampler = Sampler() if sampler is None else sampler
        self._replay: dict | None = None

    def _in_place(self) -> bool:
        """ONE shuffled batch of the whole buffer (the preset's ``AutoMiniBatchSampler()``: 1 epoch x 1 minibatch,
        stats.py:29-32): the statistics are permutation-invariant means, so the pass can read the buffer in place."""
        from cusrl_amd.sampler.mini_batch_sampler import AutoMiniBatchSampler, MiniBatchSampler

        sampler = self.sampler
        return (isinstance(sampler, (MiniBatchSampler, AutoMiniBatchSampler)) and sampler.num_epochs == 1
                and sampler.num_mini_batches in (1, (1,)) and sampler.lazy)

    def _batches(self, buffer):
        """The hook's sampler — except that ONE shuffled batch of the whole buffer (the preset's
        ``AutoMiniBatchSampler()``: 1 epoch x 1 minibatch, stats.py:29-32) is not gathered at all: the three statistics
        are means over all samples, i.e. invariant under the permutation, so the pass reads the buffer leaves in place
        (flattened views) and the 110 MB whole-buffer gather disappears.  The permutation is still drawn: the reference
        consumes one ``randperm`` from the global generator here, and the following iterations' index streams stay
        bit-identical only if this one is consumed too."""
        sampler = self.sampler
        if self._in_place():
            for metadata, _indices in sampler.iter_indices(buffer):
                if metadata["temporal"]:
                    yield metadata, buffer.sample(lambda _name, tensor: tensor)
                else:
                    yield metadata, buffer.sample(lambda _name, tensor: tensor.flatten(0, 1))
            return
        yield from sampler(buffer)

    @torch.no_grad()
    def post_update(self):
        agent = self.agent
        if self._post_update_replayed():
            return
        for _, batch in self._batches(agent.buffer):
            with agent.autocast():
                updated, _ = agent.actor(batch["observation"], memory=batch.get("actor_memory"), done=batch["done"])
            if self._gaussian_on_device(batch["action_dist"], updated):
                self._record_fused(batch, updated)
            elif self._categorical_on_device(batch["action_dist"], updated):
                self._record_fused_categorical(batch, updated)
            else:  # other policy families, CPU agents, autocast dtypes: the actor's own compute_* methods
                self._record_generic(batch, updated)

    # ------------------------------------------------------------------ compile=True: the pass from one hipGraph
    def _post_update_replayed(self) -> bool:
        """The in-place pass of a feed-forward actor is shape-static: the actor over ``[T*N]`` rows plus the one-launch
        reduction replay from a hipGraph (eagerly: ~15 launches for ~0.1 ms of device work).  The three means land in
        a persistent device tensor and are recorded from there."""
        agent, buffer = self.agent, self.agent.buffer
        actor = agent.actor
        if not (getattr(agent, "compile", False) and getattr(agent, "_graph_stream", None) is not None and self._in_place()
                and not actor.is_recurrent and agent.device.type == "cuda" and not agent.autocast_enabled):
            return False
        family = "normal" if getattr(actor.distribution, "is_normal", False) else (
            "categorical" if getattr(actor.distribution, "is_categorical", False) else None)
        if family is None or "advantage" not in buffer.storage:
            return False
        from cusrl_amd.template.graphs import GraphedRegion

        # The reference draws one permutation here (stats.py:29-32 -> mini_batch_sampler.py:56): so do we — the generator must see it
        # — but the in-place pass never reads its values, so the draw's dozen launches (~80 us of sort) go to the samplers'
        # draw-ahead stream instead of sitting between the last minibatch step and this pass.  (The per-slot record is current
        # here: `prepare_sampling` inside the iteration launches nothing; it is asked first, on this stream, should it ever.)
        from cusrl_amd.sampler.mini_batch_sampler import _prefetch_stream

        buffer.prepare_sampling(self.sampler.hot_fields if getattr(self.sampler, "lazy", False) else None)
        metadata = None
        with torch.cuda.stream(_prefetch_stream(agent.device)):
            for metadata, _indices in self.sampler.iter_indices(buffer):
                pass
        if metadata is None or metadata["temporal"]:
            return False
        flat = buffer.sample(lambda _name, tensor: tensor.flatten(0, 1))
        behaviour = flat["action_dist"]
        leaves = [flat["observation"], flat["done"], flat["action"], flat["action_logp"], flat["advantage"], *behaviour.values()]
        if any(t.dtype not in (torch.float32, torch.bool) for t in leaves):
            return False
        key = (buffer.layout_version, family, tuple(t.data_ptr() for t in leaves))
        replay = self._replay
        if replay is None or replay["key"] != key:  # the region closes over these very views

            def region():
                updated, _ = actor(flat["observation"], memory=None, done=flat["done"])
                if family == "normal":
                    return ops.policy_stats(behaviour["mean"], behaviour["std"], updated["mean"], updated["std"],
                                            flat["action"], flat["action_logp"], flat["advantage"])
                return ops.categorical_policy_stats(behaviour["logits"], updated["logits"], flat["action"],
                                                    flat["action_logp"], flat["advantage"])

            replay = self._replay = {"key": key, "region": GraphedRegion(agent, region)}
        kl, weighted_advantage, std = replay["region"].run(family).unbind(0)
        rows, metrics = flat["action_logp"].numel(), agent.metrics
        metrics.record_reduced("kl_divergence", kl, rows)
        metrics.record_reduced("importance_weighted_advantage", weighted_advantage, flat["advantage"].numel())
        if family == "normal":
            metrics.record_reduced("action_std", std, flat["action"].numel())
        return True

    # ------------------------------------------------------------------ Gaussian policy: one launch
    def _gaussian_on_device(self, behaviour, updated) -> bool:
        if not getattr(self.agent.actor.distribution, "is_normal", False):
            return False
        tensors = (behaviour.get("mean"), behaviour.get("std"), updated.get("mean"), updated.get("std"))
        return all(isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 for t in tensors)

    def _record_fused(self, batch, updated):
        behaviour, advantage = batch["action_dist"], batch["advantage"]
        kl, weighted_advantage, std = ops.policy_stats(
            behaviour["mean"], behaviour["std"], updated["mean"], updated["std"], batch["action"], batch["action_logp"],
            advantage).unbind(0)
        rows = batch["action_logp"].numel()
        metrics = self.agent.metrics
        metrics.record_reduced("kl_divergence", kl, rows)
        metrics.record_reduced("importance_weighted_advantage", weighted_advantage, advantage.numel())
        metrics.record_reduced("action_std", std, updated["std"].numel())

    # ------------------------------------------------------------------ one-hot categorical policy: one launch
    def _categorical_on_device(self, behaviour, updated) -> bool:
        if not getattr(self.agent.actor.distribution, "is_categorical", False):
            return False
        tensors = (behaviour.get("logits"), updated.get("logits"))
        return all(isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 for t in tensors)

    def _record_fused_categorical(self, batch, updated):
        advantage = batch["advantage"]
        kl, weighted_advantage, _ = ops.categorical_policy_stats(
            batch["action_dist"]["logits"], updated["logits"], batch["action"], batch["action_logp"], advantage).unbind(0)
        metrics = self.agent.metrics
        metrics.record_reduced("kl_divergence", kl, batch["action_logp"].numel())
        metrics.record_reduced("importance_weighted_advantage", weighted_advantage, advantage.numel())

    # ------------------------------------------------------------------ any other policy family
    def _record_generic(self, batch, updated):
        actor, record = self.agent.actor, self.agent.record
        record(kl_divergence=actor.compute_kl_div(batch["action_dist"], updated))
        log_ratio = actor.compute_logp(updated, batch["action"]) - batch["action_logp"]
        record(importance_weighted_advantage=batch["advantage"] * log_ratio.exp())
        if "std" in updated:
            record(action_std=updated["std"])
