"""Action distributions (counterpart of cusrl/nn/module/distribution.py:33-366).

``NormalDist`` = fp32 mean head + state-independent std vector (through a bijector), ``OneHotCategoricalDist`` =
logits head.  Log-prob / entropy / KL are written out explicitly in fp32 with the same formulas
``torch.distributions.Normal`` uses (the reference calls those, ``:195-218``); during the PPO update the fused HIP
kernel computes log-prob, entropy and their gradients instead (cusrl_amd/csrc/ppo_loss.hip).
"""

from __future__ import annotations

import math
from dataclasses import dataclass

import torch
from torch import Tensor, distributions, nn
from torch.nn.functional import one_hot

from cusrl_amd.nn.module import LinearFp32, Module, ModuleFactory, disable_autocast

__all__ = ["AdaptiveNormalDist", "Distribution", "NormalDist", "OneHotCategoricalDist", "make_bijector"]

_LOG_SQRT_2PI = math.log(math.sqrt(2 * math.pi))
_ENTROPY_CONST = 0.5 + 0.5 * math.log(2 * math.pi)


# ----------------------------------------------------------------------------------------------- bijectors
class Bijector(nn.Module):
    def forward(self, input):
        raise NotImplementedError

    def inverse(self, input):
        raise NotImplementedError


class IdentityBijector(Bijector):
    def forward(self, input):
        return input

    def inverse(self, input):
        return input


class ExponentialBijector(Bijector):
    def __init__(self, min_value: float = 0.01, max_value: float = 1.0):
        super().__init__()
        self.min_value, self.max_value = min_value, max_value
        self.min_input, self.max_input = math.log(min_value), math.log(max_value)

    def forward(self, input):
        if isinstance(input, Tensor):
            return torch.exp(input.clamp(self.min_input, self.max_input))
        return math.exp(min(max(input, self.min_input), self.max_input))

    def inverse(self, input):
        if isinstance(input, Tensor):
            return torch.log(input.clamp(self.min_value, self.max_value))
        return math.log(min(max(input, self.min_value), self.max_value))


class SoftplusBijector(Bijector):
    def forward(self, input):
        if isinstance(input, Tensor):
            return nn.functional.softplus(input)
        return math.log1p(math.exp(input))

    def inverse(self, input):
        if isinstance(input, Tensor):
            return input + torch.log(-torch.expm1(-input))
        return input + math.log(-math.expm1(-input))


def make_bijector(spec) -> Bijector:
    if spec is None:
        return IdentityBijector()
    if isinstance(spec, Bijector):
        return spec
    name, _, params = str(spec).partition("_")
    args = [float(p) for p in params.split("_") if p]
    table = {"identity": IdentityBijector, "exp": ExponentialBijector, "softplus": SoftplusBijector}
    if name not in table:
        raise ValueError(f"Unknown bijector '{spec}'")
    return table[name](*args)


# ----------------------------------------------------------------------------------------------- distributions
class DistributionFactory(ModuleFactory):
    pass


class Distribution(Module):
    Factory = DistributionFactory

    def __init__(self, input_dim: int, output_dim: int):
        super().__init__(input_dim, output_dim)
        self.mean_head = LinearFp32(input_dim, output_dim)

    def sample(self, backbone_feat: Tensor, **kwargs):
        dist_params = self(backbone_feat, **kwargs)
        return dist_params, self.sample_from_dist(dist_params)

    def sample_from_dist(self, dist_params):
        raise NotImplementedError

    def compute_logp(self, dist_params, sample: Tensor) -> Tensor:
        raise NotImplementedError

    def compute_entropy(self, dist_params) -> Tensor:
        return -self.sample_from_dist(dist_params)[1]

    def compute_kl_div(self, dist_params1, dist_params2) -> Tensor:
        sample, logp = self.sample_from_dist(dist_params1)
        return logp - self.compute_logp(dist_params2, sample)

    def determine(self, backbone_feat: Tensor, **kwargs) -> Tensor:
        return self.mean_head(backbone_feat)


class _Normal(Distribution):
    """Diagonal Gaussian over ``{"mean", "std"}`` parameter dicts; reductions keep a trailing size-1 dim."""

    is_normal = True

    @staticmethod
    def _ms(dist_params):
        return dist_params["mean"].float(), dist_params["std"].float()

    def sample_from_dist(self, dist_params):
        mean, std = self._ms(dist_params)
        if mean.is_cuda and not torch.is_grad_enabled():
            # acting path: eps from torch's generator (the reference's random stream), then ONE HIP launch for
            # action = mean + eps * std and its log-prob instead of ~12 elementwise / reduction launches
            from cusrl_amd import ops

            eps = torch.empty(mean.shape, dtype=mean.dtype, device=mean.device).normal_()
            vector = getattr(dist_params["std"], "_cusrl_row_vector", None)
            if vector is not None and vector.dtype == torch.float32 and mean.dim() == 2:
                # a state-independent std arrives as a stride-0 view of its [A] vector (StddevVector.forward): the launch
                # broadcasts it and writes the repeated [B, A] matrix the transition carries on to the rollout buffer
                action, logp, repeated = ops.normal_sample_logp(mean, vector, eps, repeat_std=True)
                dist_params["std"] = repeated
                return action, logp
            if not std.is_contiguous():  # (a view nobody resolved: the transition must carry a real [B, A] leaf)
                std = dist_params["std"] = std.contiguous()
            return ops.normal_sample_logp(mean, std, eps)
        with disable_autocast(mean.device.type):
            # same draw as Normal.rsample(): mean + std * N(0, 1) from the global generator of mean's device
            sample = mean + torch.empty(mean.shape, dtype=mean.dtype, device=mean.device).normal_() * std
            return sample, self._logp(mean, std, sample)

    @staticmethod
    def _logp(mean, std, sample):
        term = -((sample - mean) ** 2) / (2 * std**2) - std.log() - _LOG_SQRT_2PI
        return term.sum(dim=-1, keepdim=True)

    def compute_logp(self, dist_params, sample) -> Tensor:
        mean, std = self._ms(dist_params)
        with disable_autocast(mean.device.type):
            return self._logp(mean, std, sample.float())

    def compute_entropy(self, dist_params) -> Tensor:
        _, std = self._ms(dist_params)
        with disable_autocast(std.device.type):
            return (_ENTROPY_CONST + std.log()).sum(dim=-1, keepdim=True)

    def compute_kl_div(self, dist_params1, dist_params2) -> Tensor:
        (mean_p, std_p), (mean_q, std_q) = self._ms(dist_params1), self._ms(dist_params2)
        with disable_autocast(mean_p.device.type):
            var_ratio = (std_p / std_q).pow(2)
            t1 = ((mean_p - mean_q) / std_q).pow(2)
            return (0.5 * (var_ratio + t1 - 1 - var_ratio.log())).sum(dim=-1, keepdim=True)


def _resolve_init_std(init_std: float | None) -> float:
    if init_std is None:
        return 1.0
    if init_std <= 0:
        raise ValueError("'init_std' must be positive")
    return init_std


class StddevVector(nn.Module):
    """One learnable std per action dim, broadcast to the batch (``:228-247``)."""

    def __init__(self, output_dim: int, init_std: float | None = None, bijector=None):
        super().__init__()
        self.bijector = make_bijector(bijector)
        self.param = nn.Parameter(torch.ones(output_dim) * self.bijector.inverse(_resolve_init_std(init_std)))
        self.expand_when_acting = False  # set by _Normal.sample (the one caller that resolves the view in its launch)

    def forward(self, input: Tensor):
        with disable_autocast(input.device.type):
            if torch.is_grad_enabled() and self.param.requires_grad and input.is_cuda:
                # training: the bijector runs on the [A] vector and the batch sees a stride-0 view of it (no repeat
                # launch, no [B, A] gradient for sum(0) to reduce); the fused PPO objective picks the vector up
                # through `_cusrl_row_vector` and gets d_std as an [A] vector straight from its kernel
                vector = self.bijector(self.param.float()).float()
                expanded = vector.expand(*input.shape[:-1], -1)
                expanded._cusrl_row_vector = vector
                return expanded
            if self.expand_when_acting and not torch.is_grad_enabled() and input.is_cuda and input.dim() == 2:
                # acting on the GPU (the sampling launch of _Normal.sample_from_dist takes the vector and materialises
                # the repeated matrix itself): the same stride-0 view, no repeat launch per env step
                vector = self.bijector(self.param.detach().float()).float()
                expanded = vector.expand(*input.shape[:-1], -1)
                expanded._cusrl_row_vector = vector
                return expanded
            return self.bijector(self.param.float().repeat(*input.shape[:-1], 1)).float()


@dataclass(slots=True)
class NormalDistFactory(DistributionFactory):
    init_std: float | None = None
    bijector: str | Bijector | None = None

    def __call__(self, input_dim: int | None = None, output_dim: int | None = None):
        assert input_dim is not None and output_dim is not None
        return NormalDist(input_dim, output_dim, init_std=self.init_std, bijector=self.bijector)


class NormalDist(_Normal):
    Factory = NormalDistFactory

    def __init__(self, input_dim: int, output_dim: int, init_std: float | None = None, bijector=None):
        super().__init__(input_dim, output_dim)
        self.std = StddevVector(output_dim, init_std=init_std, bijector=bijector)

    def forward(self, backbone_feat: Tensor, **kwargs):
        return {"mean": self.mean_head(backbone_feat), "std": self.std(backbone_feat)}

    def std_vector(self) -> Tensor:
        """The fp32 ``[A]`` std behind the bijector, detached (what every row of the ``std`` parameter repeats)."""
        with disable_autocast(self.std.param.device.type):
            return self.std.bijector(self.std.param.detach().float()).float()

    def sample(self, backbone_feat: Tensor, **kwargs):
        if not (backbone_feat.is_cuda and not torch.is_grad_enabled()):
            return super().sample(backbone_feat, **kwargs)
        # acting on the GPU: the std stays a view of its vector until the sampling launch, which repeats it itself
        self.std.expand_when_acting = True
        try:
            head = self.mean_head
            vector = None
            if (head.bias is not None and backbone_feat.dim() == 2 and backbone_feat.dtype == torch.float32
                    and head.weight.dtype == torch.float32 and not torch.is_autocast_enabled("cuda")):
                std = self.std(backbone_feat)
                vector = getattr(std, "_cusrl_row_vector", None)
            if vector is not None and vector.dtype == torch.float32:
                # ONE launch behind the head's bias-free GEMM: + bias, action = mean + eps * std, log-prob, and the two
                # [B, A] leaves the transition carries (finished mean, repeated std) — instead of a bias-broadcast copy
                # launch in front of the GEMM, a repeat launch for the std and the sampling launch
                from cusrl_amd import ops

                raw = torch.mm(backbone_feat, head.weight.t())
                eps = torch.empty(raw.shape, dtype=raw.dtype, device=raw.device).normal_()
                action, logp, repeated, mean = ops.normal_sample_logp(raw, vector, eps, repeat_std=True, mean_bias=head.bias)
                return {"mean": mean, "std": repeated}, (action, logp)
            dist_params = self(backbone_feat, **kwargs)
        finally:
            self.std.expand_when_acting = False
        return dist_params, self.sample_from_dist(dist_params)


@dataclass(slots=True)
class AdaptiveNormalDistFactory(DistributionFactory):
    init_std: float | None = None
    bijector: str | Bijector | None = "exp"
    backward: bool = True

    def __call__(self, input_dim: int | None = None, output_dim: int | None = None):
        assert input_dim is not None and output_dim is not None
        return AdaptiveNormalDist(input_dim, output_dim, self.init_std, self.bijector, self.backward)


class AdaptiveNormalDist(_Normal):
    """State-dependent std from a second fp32 head (``:290-319``)."""

    Factory = AdaptiveNormalDistFactory

    def __init__(self, input_dim: int, output_dim: int, init_std: float | None = None, bijector="exp", backward: bool = True):
        super().__init__(input_dim, output_dim)
        self.std_head = LinearFp32(input_dim, output_dim)
        self.bijector = make_bijector(bijector)
        self.backward = backward
        self.std_head.weight.data.zero_()
        self.std_head.bias.data[:] = self.bijector.inverse(_resolve_init_std(init_std))

    def forward(self, backbone_feat: Tensor, **kwargs):
        mean = self.mean_head(backbone_feat)
        std = self.std_head(backbone_feat if self.backward else backbone_feat.detach())
        with disable_autocast(std.device.type):
            return {"mean": mean, "std": self.bijector(std).float()}


class OneHotCategoricalDistFactory(DistributionFactory):
    def __call__(self, input_dim: int | None = None, output_dim: int | None = None):
        assert input_dim is not None and output_dim is not None
        return OneHotCategoricalDist(input_dim, output_dim)


class OneHotCategoricalDist(Distribution):
    Factory = OneHotCategoricalDistFactory
    is_normal = False
    is_categorical = True  # the fused PPO objective has a one-hot categorical form (cusrl_ppo_loss_categorical_fwd_bwd)
    # acting on a GPU draws through ONE HIP launch (below); torch.distributions' own sample path reads back to the host
    # (one_hot / argument checks) and could not be captured into a hipGraph
    capture_safe = True

    @staticmethod
    def _dist(dist_params):
        return distributions.OneHotCategoricalStraightThrough(logits=dist_params["logits"].float(), validate_args=False)

    def forward(self, backbone_feat: Tensor, **kwargs):
        return {"logits": self.mean_head(backbone_feat)}

    def determine(self, backbone_feat: Tensor, **kwargs) -> Tensor:
        logits = self.mean_head(backbone_feat)
        return one_hot(logits.argmax(dim=-1), logits.size(-1)).to(dtype=logits.dtype)

    def sample_from_dist(self, dist_params):
        logits = dist_params["logits"]
        if logits.is_cuda and not torch.is_grad_enabled():
            # acting path: the Exp(1) race variables come from torch's generator exactly as torch.multinomial would draw
            # them for one sample, then ONE HIP launch does softmax, arg-max of p / q, one-hot and log-prob
            from cusrl_amd import ops

            logits = logits.float()
            return ops.categorical_sample_logp(logits, torch.empty_like(logits).exponential_(1.0))
        with disable_autocast(dist_params["logits"].device.type):
            dist = self._dist(dist_params)
            action = dist.rsample()
            return action, dist.log_prob(action).unsqueeze(-1)

    def compute_logp(self, dist_params, sample: Tensor) -> Tensor:
        with disable_autocast(dist_params["logits"].device.type):
            return self._dist(dist_params).log_prob(sample.float()).unsqueeze(-1)

    def compute_entropy(self, dist_params) -> Tensor:
        with disable_autocast(dist_params["logits"].device.type):
            return self._dist(dist_params).entropy().unsqueeze(-1)

    def compute_kl_div(self, dist_params1, dist_params2) -> Tensor:
        with disable_autocast(dist_params1["logits"].device.type):
            return distributions.kl_divergence(self._dist(dist_params1), self._dist(dist_params2)).unsqueeze(-1)
