"""KL-driven learning-rate control (counterpart of cusrl/hook/on_policy/lr_schedule.py:19-239).

After every update the mean KL divergence between the behaviour policy and the updated one (recorded by
``OnPolicyStatistics``) is averaged across ranks and turned into a multiplicative learning-rate scale:

* :class:`ThresholdLRSchedule` — step the scale down / up by a fixed factor when the KL leaves the band
  ``[desired / threshold, desired * threshold]`` (``:155-171``);
* :class:`AdaptiveLRSchedule` — integrate ``log(kl / desired)`` and, once the integral leaves ``±threshold``, scale by
  ``exp(-clip(mean log error, ±1) * scale_factor)`` and restart the integral (``:229-239``) — the ``ppo`` preset's choice
  (preset/ppo.py:56-62).

Both support a linear warm-up of the scale and rejecting an update whose KL exceeds ``max_kl_divergence`` (the
pre-update checkpoint is restored, the scale is kept).  The scale is written to ``param_group["lr"]`` of the groups
that hold actor parameters (all groups with ``scale_all_params``); captured hipGraph steps see the change because the
flat Adam step reads the learning rate from device memory (cusrl_amd/utils/flat_optimizer.py).

:class:`MiniBatchWiseLRSchedule` (``:242-296``) is the RSL-RL style rule: the threshold decision is taken inside every
minibatch step from that minibatch's own KL, before its optimizer step.  The decision is host arithmetic on a value read
back from the device, so the hook declares its ``objective`` phase eager (``Hook.eager_phases``) and ``compile=True``
leaves the minibatch steps of an agent carrying it out of hipGraph capture.
"""

from __future__ import annotations

import copy
import math

import torch

from cusrl_amd.template.hook import Hook
from cusrl_amd.utils import distributed

__all__ = ["AdaptiveLRSchedule", "MiniBatchWiseLRSchedule", "ThresholdLRSchedule"]


class _KlDrivenSchedule(Hook):
    def __init__(self, desired_kl_divergence: float, max_kl_divergence: float | None, scale_all_params: bool,
                 warmup_iterations: int, initial_scale: float):
        if desired_kl_divergence <= 0:
            raise ValueError("'desired_kl_divergence' must be positive")
        if warmup_iterations < 0:
            raise ValueError("'warmup_iterations' must be non-negative")
        if not 0 <= initial_scale <= 1:
            raise ValueError("'initial_scale' must be within [0, 1]")
        if max_kl_divergence is not None and max_kl_divergence <= 0:
            raise ValueError("'max_kl_divergence' must be positive")
        super().__init__(training_only=True)
        self.scale_all_params = scale_all_params
        self.warmup_iterations = warmup_iterations
        self.initial_scale = initial_scale
        self.desired_kl_divergence = desired_kl_divergence
        self.max_kl_divergence = max_kl_divergence
        self.register_mutable("desired_kl_divergence")
        self.register_mutable("max_kl_divergence")
        self._lr_scale = 1.0
        self._base_lrs: list[float] = []
        self._snapshot: dict | None = None

    # ------------------------------------------------------------------ hook protocol
    def post_init(self):
        self._base_lrs = [float(group["lr"]) for group in self.agent.optimizer.param_groups]

    def pre_update(self, buffer):
        if self.max_kl_divergence is not None:  # keep what a rejected update must be rolled back to
            self._snapshot = copy.deepcopy(self.agent.state_dict())

    def post_update(self):
        kl = self.agent.metrics["kl_divergence"].mean.clone()
        distributed.reduce_mean_(kl)
        kl = kl.item()
        if self.agent.iteration >= self.warmup_iterations:
            self._react(kl)
        if self.max_kl_divergence is None:
            return
        snapshot, self._snapshot = self._snapshot, None
        rejected = kl > self.max_kl_divergence
        if rejected:
            scale = self._lr_scale
            self.agent.load_state_dict(snapshot)
            self._lr_scale = scale  # the roll-back keeps the learning-rate decision
            self._write_learning_rates()
        self.agent.record(update_rejected=float(rejected))

    def apply_schedule(self, iteration: int):
        if self.warmup_iterations <= 0 or iteration > self.warmup_iterations:
            return
        progress = min(iteration, self.warmup_iterations) / self.warmup_iterations
        self._lr_scale = self.initial_scale + (1.0 - self.initial_scale) * progress
        self._write_learning_rates()
        self.agent.record(lr_scale=self._lr_scale)

    def state_dict(self):
        return {"lr_scale": self._lr_scale}

    def load_state_dict(self, state_dict):
        self._lr_scale = state_dict["lr_scale"]

    # ------------------------------------------------------------------ internals
    def _factor(self, kl_divergence: float) -> float | None:
        raise NotImplementedError

    def _react(self, kl_divergence: float):
        factor = self._factor(kl_divergence)
        if factor is not None and factor != 1.0:
            self._lr_scale *= factor
            self._write_learning_rates()
        self.agent.record(lr_scale=self._lr_scale)

    def _write_learning_rates(self):
        for base, group in zip(self._base_lrs, self.agent.optimizer.param_groups):
            holds_actor = any(name.startswith("actor.") for name in group.get("param_names", ()))
            if self.scale_all_params or holds_actor:
                group["lr"] = base * self._lr_scale


class ThresholdLRSchedule(_KlDrivenSchedule):
    def __init__(self, desired_kl_divergence: float = 0.01, *, max_kl_divergence: float | None = None,
                 threshold: float = 1.2, scale_factor: float = 1.1, scale_all_params: bool = False,
                 warmup_iterations: int = 0, initial_scale: float = 0.0):
        super().__init__(desired_kl_divergence, max_kl_divergence, scale_all_params, warmup_iterations, initial_scale)
        if threshold <= 1:
            raise ValueError("'threshold' must be greater than 1")
        if scale_factor <= 1:
            raise ValueError("'scale_factor' must be greater than 1")
        self.threshold, self.scale_factor = threshold, scale_factor

    def _factor(self, kl_divergence: float) -> float | None:
        if kl_divergence > self.desired_kl_divergence * self.threshold:
            return 1 / self.scale_factor
        if kl_divergence < self.desired_kl_divergence / self.threshold:
            return self.scale_factor
        return None


class MiniBatchWiseLRSchedule(ThresholdLRSchedule):
    """Threshold rule applied per minibatch on that minibatch's KL (all parameter groups are scaled, no roll-back)."""

    def __init__(self, desired_kl_divergence: float = 0.01, *, threshold: float = 2.0, scale_factor: float = 1.5,
                 warmup_iterations: int = 0, initial_scale: float = 0.0):
        super().__init__(desired_kl_divergence, threshold=threshold, scale_factor=scale_factor, scale_all_params=True,
                         warmup_iterations=warmup_iterations, initial_scale=initial_scale)

    def post_init(self):
        from cusrl_amd.hook.on_policy.common import OnPolicyPreparation

        super().post_init()
        for hook in self.agent.hook:
            if isinstance(hook, OnPolicyPreparation):
                hook.calculate_kl_divergence = True  # puts batch["kl_divergence"] in front of objective()

    def post_update(self):
        pass  # every decision was already taken inside the update

    def eager_phases(self):
        return ("objective",)  # .item() below

    def objective(self, metadata, batch):
        if self.agent.iteration < self.warmup_iterations:
            return None
        with torch.no_grad():
            kl = batch["kl_divergence"].mean()
        distributed.reduce_mean_(kl)
        self._react(kl.item())
        return None


class AdaptiveLRSchedule(_KlDrivenSchedule):
    def __init__(self, desired_kl_divergence: float = 0.01, *, max_kl_divergence: float | None = None,
                 threshold: float = 1.0, scale_factor: float = 0.2, scale_all_params: bool = False,
                 warmup_iterations: int = 0, initial_scale: float = 0.0):
        super().__init__(desired_kl_divergence, max_kl_divergence, scale_all_params, warmup_iterations, initial_scale)
        if threshold <= 0:
            raise ValueError("'threshold' must be positive")
        if scale_factor <= 0:
            raise ValueError("'scale_factor' must be positive")
        self.threshold, self.scale_factor = threshold, scale_factor
        self._log_error_sum, self._samples = 0.0, 0

    def _factor(self, kl_divergence: float) -> float | None:
        self._log_error_sum += math.log(max(kl_divergence, 1e-5) / self.desired_kl_divergence)
        self._samples += 1
        if -self.threshold < self._log_error_sum < self.threshold:
            return None
        mean_log_error = self._log_error_sum / self._samples
        self._log_error_sum, self._samples = 0.0, 0
        return math.exp(-min(max(mean_log_error, -1.0), 1.0) * self.scale_factor)
