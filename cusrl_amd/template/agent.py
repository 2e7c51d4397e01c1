"""Agent base class and factory (counterpart of cusrl/template/agent.py:24-391): ``act`` / ``step`` / ``update``,
the update cadence counter, autocast and GradScaler flags, checkpoint state over ``MODULES`` + ``STATEFULS``."""

from __future__ import annotations

import functools
from abc import ABC, abstractmethod
from collections.abc import Iterable, Mapping
from contextlib import contextmanager
from dataclasses import dataclass
from typing import Any, Generic, TypeVar

import numpy as np
import torch

from cusrl_amd.template.environment import Environment, EnvironmentSpec
from cusrl_amd.utils import distributed
from cusrl_amd.utils.config import device as resolve_device
from cusrl_amd.utils.metrics import Metrics

__all__ = ["Agent", "AgentFactory", "AgentT"]

AgentT = TypeVar("AgentT", bound="Agent")

_DTYPES = {"float16": torch.float16, "fp16": torch.float16, "half": torch.float16, "bfloat16": torch.bfloat16,
           "bf16": torch.bfloat16, "float32": torch.float32, "fp32": torch.float32}


@dataclass(kw_only=True)
class AgentFactory(ABC, Generic[AgentT]):
    num_steps_per_update: int
    name: str = "Agent"
    device: torch.device | str | None = None
    compile: bool | str = False
    autocast: bool | None | torch.dtype | str = False

    @abstractmethod
    def __call__(self, environment_spec: EnvironmentSpec) -> AgentT: ...

    def from_environment(self, environment: Environment) -> AgentT:
        return self(environment.spec)


def preserve_io_format(act):
    """numpy observation in -> numpy action out (agent.py:373-391)."""

    @functools.wraps(act)
    def wrapped(self, observation, state=None):
        action: torch.Tensor = act(self, observation, state)
        if isinstance(observation, np.ndarray):
            result = action.cpu().numpy()
            return result.astype(observation.dtype) if np.issubdtype(result.dtype, np.floating) else result
        dtype = observation.dtype if torch.is_floating_point(action) else None
        return action.to(device=observation.device, dtype=dtype)

    return wrapped


class Agent(ABC):
    Factory = AgentFactory
    MODULES: list[str] = []
    STATEFULS: list[str] = []

    def __init__(self, environment_spec: EnvironmentSpec, num_steps_per_update: int, name: str = "Agent",
                 device=None, compile: bool | str = False, autocast: bool | None | torch.dtype | str = False):
        spec = environment_spec
        self.environment_spec = spec
        self.observation_dim = spec.observation_dim
        self.action_dim = spec.action_dim
        self.has_state = spec.state_dim is not None
        self.state_dim = spec.state_dim or spec.observation_dim
        self.parallelism = spec.num_instances
        self.num_steps_per_update = num_steps_per_update
        self.name = name
        self.device = resolve_device(device)
        self.compile = compile
        if isinstance(autocast, str):
            self.dtype, self.autocast_enabled = _DTYPES[autocast.removeprefix("torch.")], True
        elif isinstance(autocast, torch.dtype):
            self.dtype, self.autocast_enabled = autocast, True
        else:
            self.autocast_enabled = bool(autocast)
            self.dtype = torch.float16 if self.autocast_enabled else torch.float32
        self.inference_mode = False
        self.deterministic = False
        self.transition: dict[str, Any] = {}
        self.metrics = Metrics()
        self.iteration = 0
        self.step_index = 0

    @property
    def grad_scaler_enabled(self) -> bool:
        return self.autocast_enabled and self.dtype is torch.float16

    def named_parameters(self) -> Iterable[tuple[str, torch.nn.Parameter]]:
        for name in self.MODULES:
            if (module := getattr(self, name, None)) is not None:
                yield from module.named_parameters(prefix=name)

    def parameters(self):
        for _, param in self.named_parameters():
            yield param

    @abstractmethod
    def act(self, observation, state=None): ...

    @abstractmethod
    def step(self, next_observation, reward, terminated, truncated, next_state=None, **kwargs) -> bool:
        """Counts env steps; True once ``num_steps_per_update`` were collected (agent.py:210-213)."""
        if self.inference_mode:
            return False
        self.step_index += 1
        return self.step_index >= self.num_steps_per_update

    @abstractmethod
    def update(self) -> dict[str, float]:
        self.step_index = 0
        self.iteration += 1
        summary = self.metrics.summary(self.name)
        self.metrics.clear()
        return summary

    def set_inference_mode(self, mode: bool = True, deterministic: bool | None = True):
        self.inference_mode = mode
        if deterministic is not None:
            self.deterministic = mode and deterministic

    def set_iteration(self, iteration: int):
        if iteration < 0:
            raise ValueError("Iteration must be non-negative")
        self.iteration = iteration

    def to_tensor(self, value: Any) -> torch.Tensor:
        tensor = torch.as_tensor(value, device=self.device)
        return tensor.clone() if tensor is value else tensor

    def to_nested_tensor(self, value):
        if value is None:
            return None
        if isinstance(value, (tuple, list)):
            return tuple(self.to_nested_tensor(v) for v in value)
        if isinstance(value, Mapping):
            return {k: self.to_nested_tensor(v) for k, v in value.items()}
        return self.to_tensor(value)

    def setup_module(self, module):
        return module.to(device=self.device)

    def record(self, metrics: Mapping[str, Any] | None = None, /, **kwargs):
        self.metrics.record(metrics, **kwargs)

    def state_dict(self):
        return {name: part.state_dict() for name in self.MODULES + self.STATEFULS
                if (part := getattr(self, name, None)) is not None}

    def load_state_dict(self, state_dict: dict[str, Any]):
        unused = set(state_dict)
        for name in self.MODULES + self.STATEFULS:
            if (part := getattr(self, name, None)) is None:
                continue
            if (state := state_dict.get(name)) is None:
                self.warn(f"No state_dict entry was found for '{name}'")
                continue
            unused.discard(name)
            try:
                part.load_state_dict(state)
            except (RuntimeError, ValueError) as error:
                self.warn(f"Mismatched state_dict for '{name}': {error}")
        if unused:
            self.warn(f"Unused state_dict keys: {unused}.")

    @classmethod
    def warn(cls, message):
        distributed.print_rank0(f"\033[1;33mAgent: {message}\033[0m")

    @contextmanager
    def autocast(self):
        if not self.autocast_enabled:
            yield  # fp32 hot path: skip the context-manager cost entirely
            return
        with torch.autocast(device_type=self.device.type, dtype=self.dtype, enabled=True):
            yield

    def _set_training_mode(self, mode: bool = True):
        for name in self.MODULES:
            if (module := getattr(self, name, None)) is not None:
                module.train(mode)

    @contextmanager
    def _training_mode(self):
        self._set_training_mode(True)
        try:
            yield
        finally:
            self._set_training_mode(False)
