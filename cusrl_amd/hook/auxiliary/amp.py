"""Adversarial Motion Priors (counterpart of cusrl/hook/auxiliary/amp.py:16-168): a discriminator separates agent
transitions from expert transitions; its verdict becomes a style reward added to every env step, and the buffer carries
the normalised ``agent_transition`` / ``expert_transition`` leaves the discriminator is trained on.  Discriminator and
running statistics stay torch / RunningMeanStd; the style-reward epilogue is one HIP launch
(``cusrl_amp_style_reward``)."""

from __future__ import annotations

import numpy as np
import torch
from torch import Tensor, nn

from cusrl_amd.nn.rms import RunningMeanStd
from cusrl_amd.template.hook import Hook
from cusrl_amd.utils.misc import get_first

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


class AdversarialMotionPrior(Hook):
    objective_draws_random = True  # torch.randint for the discriminator batch

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

    @torch.no_grad()
    def post_step(self, transition):
        agent_transition = transition.pop("amp_obs", None)
        if agent_transition is None:
            if self.state_indices is None:
                raise ValueError("AMP observations were not provided, and 'state_indices' is not set")
            state = get_first(transition, "state", "observation")[..., self.state_indices]
            next_state = get_first(transition, "next_state", "next_observation")[..., self.state_indices]
            agent_transition = torch.cat([state, next_state], dim=-1)
        expert_transition = self._sample_demonstration(agent_transition.size(0))
        self.transition_rms.update(agent_transition)
        self.transition_rms.update(expert_transition)
        agent_transition = self.transition_rms.normalize(agent_transition)
        transition["agent_transition"] = agent_transition
        transition["expert_transition"] = self.transition_rms.normalize(expert_transition)
        logit = self.discriminator(agent_transition)
        reward = transition["reward"]
        if reward.is_cuda:
            # on the GPU the style-reward epilogue is ALWAYS the HIP kernel; a reward it cannot update in place (strided,
            # several channels) gets the kernel's bonus added instead of a silent torch-op evaluation
            from cusrl_amd import ops

            if reward.is_contiguous() and reward.shape == logit.shape and reward.dtype == torch.float32:
                style_reward = ops.amp_style_reward_(reward, logit, self.reward_scale)
            else:
                style_reward = ops.amp_style_reward_(torch.zeros_like(logit, dtype=torch.float32), logit.float(), self.reward_scale)
                reward.add_(style_reward.to(reward.dtype))
        else:  # CPU agents (host-logic tests, no GPU in the process): the reference's torch ops
            style_reward = self.reward_scale * -torch.log(torch.clamp(1 - 1 / (1 + torch.exp(-logit)), min=1e-4))
            reward.add_(style_reward)
        self.agent.record(amp_reward=style_reward)

    def objective(self, metadata, batch):
        agent_transition = batch["agent_transition"].flatten(0, -2)
        expert_transition = batch["expert_transition"].flatten(0, -2)
        if self.batch_size is not None:
            indices = torch.randint(agent_transition.size(0), (self.batch_size,), device=self.agent.device)
            agent_transition, expert_transition = agent_transition[indices], expert_transition[indices]
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

    def _sample_demonstration(self, num_samples: int) -> Tensor:
        if self.dataset is not None:
            indices = torch.randint(self.dataset.size(0), (num_samples,), device=self.agent.device)
            return self.dataset[indices]
        sampler = self.agent.environment_spec.demonstration_sampler
        if sampler is None:
            raise ValueError("Provide either 'dataset_source' or 'environment_spec.demonstration_sampler'")
        return self.agent.to_tensor(sampler(num_samples))
