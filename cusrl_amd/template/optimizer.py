"""Optimizer construction from named parameters (counterpart of cusrl/template/optimizer.py:94-251).

Every parameter group carries ``param_names`` next to ``params`` — ``GradientClipping`` reads them
(gradient_clipping.py:64-70).  Group rules are ``(prefix | callable, options)`` pairs, first match wins.
"""

from __future__ import annotations

from collections.abc import Callable, Iterable, Mapping, Sequence
from typing import Any

import torch
from torch import nn
from torch.optim import Optimizer

__all__ = ["OptimizerCollection", "OptimizerFactory", "build_optimizer"]

Selector = Callable[[str, nn.Parameter], bool]


def _prefix_match(name: str, prefix: str) -> bool:
    return name == prefix or name.startswith(prefix + ".")


class OptimizerFactory:
    def __init__(
        self,
        cls: str | type[Optimizer],
        defaults: dict[str, Any] | None = None,
        group_overrides: Sequence[tuple[str | Selector, dict[str, Any]]] | None = None,
        param_filter: str | Sequence[str] | Selector | None = None,
    ):
        self.cls = cls
        self.defaults = dict(defaults or {})
        self.group_overrides = tuple(group_overrides or ())
        for rule in self.group_overrides:
            if not (isinstance(rule, tuple) and len(rule) == 2):
                raise TypeError("Optimizer group overrides must be (selector, options) tuples")
            selector, options = rule
            if isinstance(selector, str):
                if not selector:
                    raise ValueError("Empty prefixes are not allowed in optimizer group overrides")
            elif not callable(selector):
                raise TypeError("Group override selector must be a parameter-name prefix string or a callable")
            if not isinstance(options, dict):
                raise TypeError("Group override options must be a dict")
        if param_filter is None or callable(param_filter):
            self.param_filter = param_filter
        else:
            prefixes = (param_filter,) if isinstance(param_filter, str) else tuple(param_filter)
            for prefix in prefixes:
                if not isinstance(prefix, str):
                    raise TypeError("'param_filter' prefixes must be strings")
                if not prefix:
                    raise ValueError("Empty prefixes are not allowed in 'param_filter'")
            self.param_filter = prefixes

    def _selected(self, name: str, param: nn.Parameter) -> bool:
        if self.param_filter is None:
            return True
        if callable(self.param_filter):
            return bool(self.param_filter(name, param))
        return any(_prefix_match(name, prefix) for prefix in self.param_filter)

    def _rule_for(self, name: str, param: nn.Parameter) -> tuple[int, dict[str, Any]]:
        for index, (selector, options) in enumerate(self.group_overrides):
            if _prefix_match(name, selector) if isinstance(selector, str) else selector(name, param):
                return index, options
        return -1, {}

    def __call__(self, named_parameters: Iterable[tuple[str, nn.Parameter]]) -> Optimizer:
        optim_cls = getattr(torch.optim, self.cls) if isinstance(self.cls, str) else self.cls
        groups: dict[int, dict[str, Any]] = {}
        for name, param in named_parameters:
            if not param.requires_grad or not self._selected(name, param):
                continue
            index, options = self._rule_for(name, param)
            group = groups.setdefault(index, {"param_names": [], "params": [], **options})
            group["param_names"].append(name)
            group["params"].append(param)
        if not groups:
            raise ValueError("No trainable parameters matched the optimizer filter")
        return optim_cls(list(groups.values()), **self.defaults)


class OptimizerCollection:
    """Several named optimizers behind the subset of the Optimizer interface the agent uses."""

    def __init__(self, optimizers: Mapping[str, Optimizer]):
        if not optimizers:
            raise ValueError("At least one optimizer is required")
        self.optimizers = dict(optimizers)
        seen: set[int] = set()
        for name, optimizer in self.optimizers.items():
            if not isinstance(name, str) or not name:
                raise ValueError("Optimizer names must be non-empty strings")
            for group in optimizer.param_groups:
                group["optimizer_name"] = name
                for param in group["params"]:
                    if id(param) in seen:
                        raise ValueError("Parameter is assigned to multiple optimizers")
                    seen.add(id(param))

    @property
    def param_groups(self) -> list[dict[str, Any]]:
        return [group for optimizer in self.optimizers.values() for group in optimizer.param_groups]

    def zero_grad(self, *args, **kwargs):
        for optimizer in self.optimizers.values():
            optimizer.zero_grad(*args, **kwargs)

    def step(self, *args, **kwargs):
        for optimizer in self.optimizers.values():
            optimizer.step(*args, **kwargs)

    def state_dict(self):
        return {name: optimizer.state_dict() for name, optimizer in self.optimizers.items()}

    def load_state_dict(self, state_dict: Mapping[str, Any]):
        if set(state_dict) != set(self.optimizers):
            raise ValueError(f"Mismatched optimizer collection state_dict keys: {sorted(state_dict)} vs {sorted(self.optimizers)}")
        for name, optimizer in self.optimizers.items():
            optimizer.load_state_dict(state_dict[name])


def build_optimizer(factory: OptimizerFactory | Mapping[str, OptimizerFactory], named_parameters):
    """One optimizer, or a named collection where each factory takes what the previous ones left."""
    remaining = tuple((n, p) for n, p in named_parameters if p.requires_grad)

    def left_after(optimizer):
        taken = {id(p) for group in optimizer.param_groups for p in group["params"]}
        return tuple((n, p) for n, p in remaining if id(p) not in taken)

    if isinstance(factory, OptimizerFactory):
        optimizer = factory(remaining)
        remaining = left_after(optimizer)
        result: Any = optimizer
    else:
        optimizers = {}
        for name, sub_factory in factory.items():
            if not isinstance(name, str) or not name:
                raise ValueError("Optimizer names must be non-empty strings")
            optimizers[name] = sub_factory(remaining)
            remaining = left_after(optimizers[name])
        result = OptimizerCollection(optimizers)
    if remaining:
        raise ValueError(f"Trainable parameters were not assigned to any optimizer: {[n for n, _ in remaining]!r}")
    return result
