from cusrl_amd.nn.actor import Actor, Value
from cusrl_amd.nn.distribution import AdaptiveNormalDist, Distribution, NormalDist, OneHotCategoricalDist
from cusrl_amd.nn.module import LinearFp32, Mlp, Module, ModuleFactory
from cusrl_amd.nn.rms import RunningMeanStd
from cusrl_amd.nn.rnn import Gru, Lstm, Rnn

__all__ = [
    "Actor",
    "AdaptiveNormalDist",
    "Distribution",
    "LinearFp32",
    "Mlp",
    "Module",
    "ModuleFactory",
    "NormalDist",
    "OneHotCategoricalDist",
    "Gru",
    "Lstm",
    "Rnn",
    "RunningMeanStd",
    "Value",
]
