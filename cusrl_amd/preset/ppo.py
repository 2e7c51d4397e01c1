"""The `ppo` preset (counterpart of cusrl/preset/ppo.py:19-182, optimizer.py:9-23): same hook order — it is
semantics: value target -> GAE -> advantage normalisation in ``pre_update``; value loss -> policy evaluation ->
surrogate -> entropy in ``objective``; clipping in ``pre_optim``; statistics in ``post_update`` — same field
names and defaults, so user code composing ``PpoAgentFactory`` / ``ppo_hook_suite`` is unchanged."""

from __future__ import annotations

from collections.abc import Sequence
from dataclasses import dataclass, field
from typing import Any

import torch

from cusrl_amd import hook as hooks
from cusrl_amd.nn import Actor, Mlp, NormalDist, OneHotCategoricalDist, Rnn, Value
from cusrl_amd.sampler import AutoMiniBatchSampler
from cusrl_amd.template.actor_critic import ActorCritic, ActorCriticFactory
from cusrl_amd.template.agent import AgentFactory
from cusrl_amd.template.environment import EnvironmentSpec
from cusrl_amd.template.hook import Hook
from cusrl_amd.template.optimizer import OptimizerFactory

__all__ = ["AdamFactory", "AmpAgentFactory", "PpoAgentFactory", "RecurrentPpoAgentFactory", "ppo_hook_suite"]


class AdamFactory(OptimizerFactory):
    def __init__(self, defaults: dict[str, Any] | None = None, group_overrides=None, param_filter=None):
        super().__init__("Adam", defaults=defaults, group_overrides=group_overrides, param_filter=param_filter)


def ppo_hook_suite(
    orthogonal_init: bool = True,
    normalize_observation: bool = False,
    gae_gamma: float = 0.99,
    gae_lamda: float = 0.95,
    gae_lamda_value: float | None = None,
    normalize_advantage: bool = True,
    value_loss_weight: float = 0.5,
    value_loss_clip: float | None = None,
    surrogate_clip_ratio: float = 0.2,
    surrogate_loss_weight: float = 1.0,
    entropy_loss_weight: float = 0.01,
    max_grad_norm: float | None = 1.0,
    grad_clip_groups: dict[str, float] | None = None,
    desired_kl_divergence: float | None = None,
    max_kl_divergence: float | None = None,
    empty_cuda_cache: bool = False,
) -> list[Hook]:
    suite = [
        hooks.ModuleInitialization(init_actor=orthogonal_init, init_critic=orthogonal_init),
        hooks.ObservationNormalization() if normalize_observation else None,
        hooks.ValueComputation(),
        hooks.GeneralizedAdvantageEstimation(gamma=gae_gamma, lamda=gae_lamda, lamda_value=gae_lamda_value),
        hooks.AdvantageNormalization() if normalize_advantage else None,
        hooks.ValueLoss(weight=value_loss_weight, loss_clip=value_loss_clip),
        hooks.OnPolicyPreparation(),
        hooks.PpoSurrogateLoss(clip_ratio=surrogate_clip_ratio, weight=surrogate_loss_weight),
        hooks.EntropyLoss(weight=entropy_loss_weight),
        hooks.GradientClipping(max_grad_norm, grad_clip_groups),
        hooks.OnPolicyStatistics(sampler=AutoMiniBatchSampler()),
        (hooks.AdaptiveLRSchedule(desired_kl_divergence, max_kl_divergence=max_kl_divergence)
         if desired_kl_divergence is not None else None),
        hooks.EmptyCudaCache() if empty_cuda_cache else None,
    ]
    return [h for h in suite if h is not None]


def get_distribution_factory(action_space_type: str, **kwargs):
    if action_space_type == "continuous":
        return NormalDist.Factory(**kwargs)
    if action_space_type == "discrete":
        return OneHotCategoricalDist.Factory()
    raise ValueError(f"Unsupported action space type '{action_space_type}'")


@dataclass(kw_only=True)
class PpoAgentFactory(AgentFactory):
    num_steps_per_update: int = 24
    actor_hidden_dims: Sequence[int] = (256, 128)
    critic_hidden_dims: Sequence[int] = (256, 128)
    activation_fn: str | type[torch.nn.Module] = "ReLU"
    action_space_type: str = "continuous"
    lr: float = 2e-4
    sampler_epochs: int = 5
    sampler_mini_batches: int = 4
    orthogonal_init: bool = True
    init_distribution_std: float | None = None
    normalize_observation: bool = False
    gae_gamma: float = 0.99
    gae_lamda: float = 0.95
    gae_lamda_value: float | None = None
    normalize_advantage: bool = True
    value_loss_weight: float = 0.5
    value_loss_clip: float | None = None
    surrogate_clip_ratio: float = 0.2
    surrogate_loss_weight: float = 1.0
    entropy_loss_weight: float = 0.01
    max_grad_norm: float | None = 1.0
    grad_clip_groups: dict[str, float] = field(default_factory=dict)
    desired_kl_divergence: float | None = None
    max_kl_divergence: float | None = None
    optimizer_kwargs: dict[str, Any] = field(default_factory=dict)
    """Extra torch.optim.Adam kwargs (extension), e.g. ``{"fused": True}`` for the single-kernel Adam."""
    empty_cuda_cache: bool = False
    """Release the allocator's cached blocks after every update (``hooks.EmptyCudaCache``; the reference has this field on
    the recurrent factory only, preset/ppo.py:243, where it defaults to True)."""

    def to_underlying(self) -> ActorCriticFactory:
        def backbone(dims):
            return Mlp.Factory(hidden_dims=dims, activation_fn=self.activation_fn, ends_with_activation=True)

        return ActorCriticFactory(
            num_steps_per_update=self.num_steps_per_update,
            actor_factory=Actor.Factory(
                backbone_factory=backbone(self.actor_hidden_dims),
                distribution_factory=get_distribution_factory(self.action_space_type, init_std=self.init_distribution_std),
            ),
            critic_factory=Value.Factory(backbone_factory=backbone(self.critic_hidden_dims)),
            optimizer_factory=AdamFactory(defaults={
                "lr": self.lr, **({"capturable": True, "fused": True} if self.compile else {}), **self.optimizer_kwargs}),
            sampler=AutoMiniBatchSampler(num_epochs=self.sampler_epochs, num_mini_batches=self.sampler_mini_batches),
            hooks=ppo_hook_suite(
                orthogonal_init=self.orthogonal_init,
                normalize_observation=self.normalize_observation,
                gae_gamma=self.gae_gamma,
                gae_lamda=self.gae_lamda,
                gae_lamda_value=self.gae_lamda_value,
                normalize_advantage=self.normalize_advantage,
                value_loss_weight=self.value_loss_weight,
                value_loss_clip=self.value_loss_clip,
                surrogate_clip_ratio=self.surrogate_clip_ratio,
                surrogate_loss_weight=self.surrogate_loss_weight,
                entropy_loss_weight=self.entropy_loss_weight,
                max_grad_norm=self.max_grad_norm,
                grad_clip_groups=self.grad_clip_groups,
                desired_kl_divergence=self.desired_kl_divergence,
                max_kl_divergence=self.max_kl_divergence,
                empty_cuda_cache=self.empty_cuda_cache,
            ),
            name=self.name,
            device=self.device,
            compile=self.compile,
            autocast=self.autocast,
        )

    def __call__(self, environment_spec: EnvironmentSpec) -> ActorCritic:
        return self.to_underlying()(environment_spec)


@dataclass(kw_only=True)
class RecurrentPpoAgentFactory(PpoAgentFactory):
    """PPO with GRU / LSTM actor and critic (counterpart of cusrl/preset/ppo.py:185-298): same hook suite; the sampler
    switches to whole-sequence (temporal) minibatches because the buffer holds ``*_memory`` leaves."""

    rnn_type: str = "LSTM"
    actor_num_layers: int = 2
    actor_hidden_size: int = 256
    critic_num_layers: int = 2
    critic_hidden_size: int = 256
    empty_cuda_cache: bool = True  # preset/ppo.py:243

    def to_underlying(self) -> ActorCriticFactory:
        underlying = super().to_underlying()
        underlying.actor_factory = Actor.Factory(
            backbone_factory=Rnn.Factory(self.rnn_type, num_layers=self.actor_num_layers, hidden_size=self.actor_hidden_size),
            distribution_factory=get_distribution_factory(self.action_space_type, init_std=self.init_distribution_std),
        )
        underlying.critic_factory = Value.Factory(
            backbone_factory=Rnn.Factory(self.rnn_type, num_layers=self.critic_num_layers, hidden_size=self.critic_hidden_size))
        return underlying


@dataclass(kw_only=True)
class AmpAgentFactory(PpoAgentFactory):
    """PPO + Adversarial Motion Priors (counterpart of cusrl/preset/amp.py:12-53): reward shaping and the AMP hook are
    inserted before ``value_computation`` so the style reward is in the buffer when the value target is built."""

    extrinsic_reward_scale: float = 1.0
    amp_discriminator_hidden_dims: Sequence[int] = (256, 128)
    amp_dataset_source: Any = None
    amp_state_indices: Any = None
    amp_batch_size: int = 512
    amp_reward_scale: float = 1.0
    amp_loss_weight: float = 1.0
    amp_grad_penalty_weight: float = 5.0

    def to_underlying(self) -> ActorCriticFactory:
        underlying = super().to_underlying()
        underlying.register_hook(hooks.RewardShaping(scale=self.extrinsic_reward_scale), before="value_computation")
        underlying.register_hook(
            hooks.AdversarialMotionPrior(
                discriminator_factory=Mlp.Factory(hidden_dims=self.amp_discriminator_hidden_dims, activation_fn=self.activation_fn),
                dataset_source=self.amp_dataset_source, state_indices=self.amp_state_indices, batch_size=self.amp_batch_size,
                reward_scale=self.amp_reward_scale, loss_weight=self.amp_loss_weight,
                grad_penalty_weight=self.amp_grad_penalty_weight),
            after="reward_shaping",
        )
        return underlying
