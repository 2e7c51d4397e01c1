from cusrl_amd.nn.actor import Actor, Value
from cusrl_amd.nn.distribution import AdaptiveNormalDist, Distribution, NormalDist, OneHotCategoricalDist
from cusrl_amd.nn.module import LinearFp32, Mlp, Module, ModuleFactory

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
    "Value",
]
