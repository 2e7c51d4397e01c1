"""Online observation / state normalisation (counterpart of cusrl/hook/mdp/observation.py:59-255).

``pre_act`` and ``post_step`` update running statistics with the incoming (next) observation and replace it by its
normalised, clamped value, keeping the raw tensors under ``original_*`` (they become extra buffer leaves and ride
the same push / gather launches).  On MI355X an update is masked column statistics + merge + normalise = three HIP
launches with the sample count kept on the device; the reference's ``observation[last_done]`` boolean-mask select
(observation.py:206-208), which synchronises with the host every step, becomes the kernel's row mask.
"""

from __future__ import annotations

import numpy as np
import torch
from torch import Tensor

from cusrl_amd.nn.rms import RunningMeanStd, mean_var_count
from cusrl_amd.template.hook import Hook

__all__ = ["ObservationNormalization"]


class ObservationNormalization(Hook):
    def __init__(self, max_count: int | None = None, defer_synchronization: bool = False, renormalize: bool = False):
        if max_count is not None and max_count <= 0:
            raise ValueError("'max_count' must be positive or None")
        super().__init__()
        self.max_count = max_count
        self.defer_synchronization = defer_synchronization
        self.renormalize = renormalize
        self.frozen: bool = False
        self.register_mutable("frozen")
        self.observation_rms: RunningMeanStd
        self.state_rms: RunningMeanStd | None = None
        self._observation_is_subset_of_state = None
        self._last_done: Tensor | None = None

    def freeze(self):
        self.frozen = True
        return self

    def init(self):
        spec = self.agent.environment_spec
        if spec.mirror_observation is not None or spec.mirror_state is not None:
            raise NotImplementedError("symmetry-aware statistics (mirror_*) are out of scope (SURVEY.md §2 row 15)")
        subset = spec.observation_is_subset_of_state
        if subset is not None:
            if not self.agent.has_state:
                raise ValueError("'observation_is_subset_of_state' is set without defining the state")
            if isinstance(subset, (np.ndarray, list, tuple)):
                subset = self.agent.to_tensor(np.asarray(subset))
            self._observation_is_subset_of_state = subset
            self.register_module("observation_rms", RunningMeanStd(self.agent.observation_dim))
        else:
            self.register_module("observation_rms", RunningMeanStd(
                self.agent.observation_dim, max_count=self.max_count, groups=spec.observation_stat_groups,
                excluded_indices=spec.observation_normalization_excluded_indices))
        if self.agent.has_state:
            self.register_module("state_rms", RunningMeanStd(
                self.agent.state_dim, max_count=self.max_count, groups=spec.state_stat_groups,
                excluded_indices=spec.state_normalization_excluded_indices))
        else:
            self.state_rms = None

    def collective_phases(self):
        # every statistics update all-gathers (mean, var, count) unless synchronisation is deferred to pre_update
        return () if (self.defer_synchronization or self.frozen) else ("act",)

    def on_replay(self, phase):
        # a replayed statistics update happened on the device only: the modules' "synchronised" flags are host state
        if self.defer_synchronization and not (self.frozen or self.agent.inference_mode):
            for rms in (self.observation_rms, self.state_rms):
                if rms is not None:
                    rms._is_synchronized = False

    # ---- rollout
    def pre_act(self, transition):
        observation, state = transition["observation"], transition.get("state")
        if self._last_done is None:
            # persistent mask buffer (all rows on the very first step): a fixed address keeps pre_act capturable
            # into the act hipGraph, post_step refreshes its contents in place
            self._last_done = torch.ones(observation.shape[0], dtype=torch.bool, device=observation.device)
            self._update_rms(observation, state, self._last_done)
        elif not self.agent.environment_spec.final_state_is_missing:
            # after the first step only freshly reset envs carry an observation the statistics have not seen yet
            self._update_rms(observation, state, self._last_done)
        transition["original_observation"] = observation
        transition["observation"] = self.observation_rms.normalize(observation)
        if self.state_rms is not None:
            transition["original_state"] = state
            transition["state"] = self.state_rms.normalize(state)

    def post_step(self, transition):
        next_observation, next_state = transition["next_observation"], transition.get("next_state")
        self._update_rms(next_observation, next_state)
        done = transition["done"].squeeze(-1)
        if self._last_done is None or self._last_done.shape != done.shape:
            self._last_done = done.clone()
        else:
            self._last_done.copy_(done)
        transition["original_next_observation"] = next_observation
        transition["next_observation"] = self.observation_rms.normalize(next_observation)
        if self.state_rms is not None:
            transition["original_next_state"] = next_state
            transition["next_state"] = self.state_rms.normalize(next_state)

    def _update_rms(self, observation: Tensor, state: Tensor | None, mask: Tensor | None = None):
        if self.agent.inference_mode or self.frozen:
            return
        synchronize = not self.defer_synchronization
        if state is not None:
            self.state_rms.update_from_stats(*mean_var_count(state, mask), synchronize=synchronize)
        if self._observation_is_subset_of_state is not None:
            self._copy_observation_stats_from_state()
        else:
            self.observation_rms.update_from_stats(*mean_var_count(observation, mask), synchronize=synchronize)

    def _copy_observation_stats_from_state(self):
        index = self._observation_is_subset_of_state
        self.observation_rms.mean.copy_(self.state_rms.mean[index])
        self.observation_rms.var.copy_(self.state_rms.var[index])
        self.observation_rms.std.copy_(self.state_rms.std[index])
        self.observation_rms._count.copy_(self.state_rms._count)

    # ---- update
    def pre_update(self, buffer):
        if self.defer_synchronization:
            if self.state_rms is not None:
                self.state_rms.synchronize()
            if self._observation_is_subset_of_state is not None:
                self._copy_observation_stats_from_state()
            else:
                self.observation_rms.synchronize()

    def objective(self, metadata, batch):
        if self.renormalize:
            batch["observation"] = self.observation_rms.normalize(batch["original_observation"])
            batch["next_observation"] = self.observation_rms.normalize(batch["original_next_observation"])
            if self.state_rms is not None:
                batch["state"] = self.state_rms.normalize(batch["original_state"])
                batch["next_state"] = self.state_rms.normalize(batch["original_next_state"])
