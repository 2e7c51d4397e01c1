"""Policy and value networks (counterparts of cusrl/nn/module/actor.py:26-274 and critic.py:27-101): a backbone
``Module`` followed by a distribution head (actor) or an fp32 value head (critic)."""

from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Callable

import torch
from torch import Tensor, nn

from cusrl_amd.nn.distribution import Distribution, NormalDist
from cusrl_amd.nn.module import Linear, Module, ModuleFactory, fused_inference_layers

__all__ = ["Actor", "Value"]


@dataclass(slots=True)
class ActorFactory(ModuleFactory):
    backbone_factory: Callable[[int | None, int | None], Module]
    distribution_factory: Callable[[int | None, int | None], Distribution]
    latent_dim: int | None = None

    def __call__(self, input_dim: int | None = None, output_dim: int | None = None):
        backbone = self.backbone_factory(input_dim, self.latent_dim)
        return Actor(backbone, self.distribution_factory(backbone.output_dim, output_dim))


class Actor(Module):
    Factory = ActorFactory

    def __init__(self, backbone: Module, distribution: Distribution):
        super().__init__(backbone.input_dim, distribution.output_dim, backbone.is_recurrent)
        self.backbone: Module = backbone.rnn_compatible()
        self.distribution: Distribution = distribution
        self.latent_dim = self.backbone.output_dim
        self.backbone_kwargs: dict[str, Any] = {}
        self.distribution_kwargs: dict[str, Any] = {}
        # the no-grad pass (acting, statistics) of an Mlp backbone + Normal head as one launch; it leaves no "backbone.output"
        # in `intermediate_repr` — a hook that reads the latent of a no-grad pass switches this off
        self.fused_inference = True
        # exploration noise drawn ahead of the act step that uses it (template/graphs.py GraphedRolloutStep: the draws of a whole
        # rollout are issued before it, off its serial chain); taken by the NEXT fused explore pass, once
        self.pending_noise: Tensor | None = None
        self.noise_shape: tuple[int, int] | None = None  # of the last fused explore pass (what an ahead-of-time draw must look like)

    def _fused_layers(self, observation, memory, backbone_kwargs, distribution_kwargs):
        if (not self.fused_inference or type(self.distribution) is not NormalDist or self.backbone_kwargs or self.distribution_kwargs
                or distribution_kwargs or any(key != "sequential" for key in (backbone_kwargs or ()))):
            return None
        return fused_inference_layers(self.backbone, self.distribution.mean_head, observation, memory)

    def clear_intermediate_repr(self):
        super().clear_intermediate_repr()
        self.backbone.clear_intermediate_repr()
        self.distribution.clear_intermediate_repr()

    def _encode(self, observation, memory, done, backbone_kwargs):
        kwargs = {**self.backbone_kwargs, **(backbone_kwargs or {})}
        if done is not None:
            kwargs["done"] = done
        latent, memory = self.backbone(observation, memory=memory, **kwargs)
        self.intermediate_repr["backbone.output"] = latent
        return latent, memory

    def forward(self, observation: Tensor, memory=None, done: Tensor | None = None, backbone_kwargs=None,
                distribution_kwargs=None, forward_type: str | None = "forward", deterministic: bool = False):
        """``forward_type``: "forward" -> (dist_params, memory); "explore" -> (dist_params, (action, logp), memory);
        "act" / "act_deterministic" -> (action, memory)  (actor.py:69-92)."""
        if forward_type == "forward":
            if (layers := self._fused_layers(observation, memory, backbone_kwargs, distribution_kwargs)) is not None:
                from cusrl_amd import ops

                self.intermediate_repr.pop("backbone.output", None)
                mean = ops.mlp2_forward(observation, layers)
                return {"mean": mean, "std": self.distribution.std(mean)}, memory
            latent, memory = self._encode(observation, memory, done, backbone_kwargs)
            dist_kwargs = {**self.distribution_kwargs, **(distribution_kwargs or {})}
            return self.distribution(latent, observation=observation, **dist_kwargs), memory
        if forward_type == "act_deterministic":
            forward_type, deterministic = "act", True
        if forward_type not in ("explore", "act"):
            raise ValueError(f"Unsupported 'forward_type' value: {forward_type!r}")
        if not deterministic and (layers := self._fused_layers(observation, memory, backbone_kwargs, distribution_kwargs)) is not None:
            # acting: backbone, mean head, Normal.rsample and its log-prob from ONE launch; eps from torch's generator exactly
            # as the unfused path draws it (distribution.py NormalDist.sample)
            from cusrl_amd import ops

            self.intermediate_repr.pop("backbone.output", None)
            vector = self.distribution.std_vector()
            shape = (observation.shape[0], vector.numel())
            eps, self.pending_noise, self.noise_shape = self.pending_noise, None, shape
            if eps is None:
                eps = torch.empty(shape, dtype=torch.float32, device=observation.device).normal_()
            elif tuple(eps.shape) != shape or eps.dtype != torch.float32 or eps.device != observation.device or not eps.is_contiguous():
                raise RuntimeError(f"exploration noise drawn ahead for another pass: {tuple(eps.shape)} vs {shape}")
            action, logp, mean, repeated = ops.mlp2_forward(observation, layers, std=vector, eps=eps)
            if forward_type == "act":
                return action, memory
            return {"mean": mean, "std": repeated}, (action, logp), memory
        latent, memory = self._encode(observation, memory, None, backbone_kwargs)
        dist_kwargs = {**self.distribution_kwargs, **(distribution_kwargs or {})}
        if deterministic:
            dist_params = self.distribution(latent, observation=observation, **dist_kwargs)
            action = self.distribution.determine(latent, observation=observation, **dist_kwargs)
            logp = self.distribution.compute_logp(dist_params, action)
        else:
            dist_params, (action, logp) = self.distribution.sample(latent, observation=observation, **dist_kwargs)
        if forward_type == "act":
            return action, memory
        return dist_params, (action, logp), memory

    def explore(self, observation, memory=None, deterministic=False, backbone_kwargs=None, distribution_kwargs=None):
        return self(observation, memory=memory, deterministic=deterministic, backbone_kwargs=backbone_kwargs,
                    distribution_kwargs=distribution_kwargs, forward_type="explore")

    def act(self, observation, memory=None, deterministic=False, backbone_kwargs=None, distribution_kwargs=None):
        return self(observation, memory=memory, deterministic=deterministic, backbone_kwargs=backbone_kwargs,
                    distribution_kwargs=distribution_kwargs, forward_type="act")

    def compute_logp(self, dist_params, action):
        return self.distribution.compute_logp(dist_params, action)

    def compute_entropy(self, dist_params):
        return self.distribution.compute_entropy(dist_params)

    def compute_kl_div(self, dist_params1, dist_params2):
        return self.distribution.compute_kl_div(dist_params1, dist_params2)

    def step_memory(self, observation, memory=None, **kwargs):
        return self.backbone.step_memory(observation, memory, **kwargs)

    def reset_memory(self, memory, done=None):
        self.backbone.reset_memory(memory, done)


@dataclass(slots=True)
class ValueFactory(ModuleFactory):
    backbone_factory: Callable[[int | None, int | None], Module]
    value_head_factory: Callable[[int, int], nn.Module] = Linear  # nn.Linear + wide-batch weight gradient
    latent_dim: int | None = None
    action_aware: bool = False

    def __call__(self, input_dim: int | None = None, output_dim: int | None = 1):
        backbone = self.backbone_factory(input_dim, self.latent_dim)
        return Value(backbone, self.value_head_factory(backbone.output_dim, output_dim), action_aware=self.action_aware)


class Value(Module):
    Factory = ValueFactory

    def __init__(self, backbone: Module, value_head: nn.Module, action_aware: bool = False):
        with torch.no_grad():
            output_dim = value_head(torch.zeros(1, backbone.output_dim)).numel()
        super().__init__(backbone.input_dim, output_dim, backbone.is_recurrent)
        self.backbone: Module = backbone.rnn_compatible()
        self.value_head = value_head
        self.action_aware = action_aware
        self.backbone_kwargs: dict[str, Any] = {}
        self.fused_inference = True  # (as for Actor)

    def forward(self, state: Tensor, *, action: Tensor | None = None, memory=None, done: Tensor | None = None, **kwargs):
        if self.action_aware:
            if action is None:
                raise ValueError("Action must be provided when 'action_aware' is True")
            state = torch.cat([state, action], dim=-1)
        if (self.fused_inference and not self.backbone_kwargs and not kwargs
                and (layers := fused_inference_layers(self.backbone, self.value_head, state, memory)) is not None):
            from cusrl_amd import ops

            self.intermediate_repr.pop("backbone.output", None)
            return ops.mlp2_forward(state, layers), memory  # the no-grad pass (value targets) as one launch
        kwargs = {**self.backbone_kwargs, **kwargs}
        if done is not None:
            kwargs["done"] = done
        latent, memory = self.backbone(state, memory=memory, **kwargs)
        self.intermediate_repr["backbone.output"] = latent
        if latent.dtype == torch.float32 and not torch.is_autocast_enabled(latent.device.type):
            return self.value_head(latent), memory
        with torch.autocast(device_type=latent.device.type, enabled=False):
            return self.value_head(latent.float()), memory  # the head always runs in fp32 (critic.py:87-88)

    def evaluate(self, state: Tensor, *, action=None, memory=None, done=None, **kwargs) -> Tensor:
        return self(state, action=action, memory=memory, done=done, **kwargs)[0]

    def clear_intermediate_repr(self):
        super().clear_intermediate_repr()
        self.backbone.clear_intermediate_repr()

    def step_memory(self, state, memory=None, **kwargs):
        return self.backbone.step_memory(state, memory, **kwargs)

    def reset_memory(self, memory, done=None):
        return self.backbone.reset_memory(memory, done=done)
