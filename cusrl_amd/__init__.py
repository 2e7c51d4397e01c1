"""cusrl_amd — MI355X-native vectorised-rollout + PPO-update engine behind the cusrl plugin surface.

``import cusrl_amd as cusrl`` and the `ppo` preset, ``Buffer`` / ``Sampler`` / ``Hook`` / ``Trainer`` read like
chengruiz/cusrl; the per-iteration hot path (buffer append, next_value, GAE, advantage normalisation, minibatch
gather, PPO objective forward+backward) runs as hand-written HIP kernels for gfx950 through ``libcusrl_hip.so``.
"""

from cusrl_amd import hook, nn, preset, sampler, template, testing, utils
from cusrl_amd.nn import (
    Actor, AdaptiveNormalDist, Distribution, Gru, LinearFp32, Lstm, Mlp, Module, ModuleFactory, NormalDist,
    OneHotCategoricalDist, Rnn, RunningMeanStd, Value,
)
from cusrl_amd.sampler import (
    AutoMiniBatchSampler, AutoRandomSampler, MiniBatchSampler, RandomSampler, TemporalMiniBatchSampler, TemporalRandomSampler,
)
from cusrl_amd.template import (
    ActorCritic,
    Agent,
    Buffer,
    Environment,
    EnvironmentSpec,
    Hook,
    OptimizerFactory,
    Sampler,
    Trainer,
    TrainerHook,
)
from cusrl_amd.utils import CONFIG as config
from cusrl_amd.utils import device, set_global_seed

__version__ = "0.1.0"

__all__ = [
    "Actor",
    "ActorCritic",
    "AdaptiveNormalDist",
    "Agent",
    "AutoMiniBatchSampler",
    "AutoRandomSampler",
    "Buffer",
    "Distribution",
    "Environment",
    "EnvironmentSpec",
    "Gru",
    "Hook",
    "LinearFp32",
    "Lstm",
    "MiniBatchSampler",
    "Mlp",
    "Module",
    "ModuleFactory",
    "NormalDist",
    "OneHotCategoricalDist",
    "OptimizerFactory",
    "RandomSampler",
    "Rnn",
    "RunningMeanStd",
    "Sampler",
    "TemporalMiniBatchSampler",
    "TemporalRandomSampler",
    "Trainer",
    "TrainerHook",
    "Value",
    "config",
    "device",
    "hook",
    "nn",
    "preset",
    "sampler",
    "set_global_seed",
    "template",
    "testing",
    "utils",
]
