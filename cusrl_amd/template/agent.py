"""Agent base class and factory — the plugin surface every hook and the trainer program against
(cusrl/template/agent.py:24-391 defines it: ``act`` / ``step`` / ``update``, the update cadence, inference / deterministic
flags, autocast + GradScaler switches, ``state_dict`` over ``MODULES`` + ``STATEFULS``).

The surface is the reference's (names, arguments, return values — SURVEY.md §8b); the organisation is this package's:

* precision is ONE value object (:class:`Precision`) resolved from the ``autocast`` argument, instead of three attributes
  assigned in branches of ``__init__`` — ``agent.dtype`` / ``autocast_enabled`` / ``grad_scaler_enabled`` read it;
* everything that walks "the named parts of the agent" — parameters, train / eval mode, checkpoint save and restore — goes
  through one generator (:meth:`Agent._parts`), and restoring a checkpoint first sorts the keys into restored / absent /
  unknown and then reports once;
* the fp32 hot path never enters ``torch.autocast`` (:meth:`Agent.autocast` yields immediately): the context manager costs
  more on the host than some of the kernels it would wrap.
"""

from __future__ import annotations

import functools
from abc import ABC, abstractmethod
from collections.abc import Iterable, Iterator, Mapping
from contextlib import contextmanager, nullcontext
from dataclasses import dataclass
from typing import Any, Generic, NamedTuple, TypeVar

import numpy as np
import torch

from cusrl_amd.template.environment import Environment, EnvironmentSpec
from cusrl_amd.utils import distributed
from cusrl_amd.utils.config import device as resolve_device
from cusrl_amd.utils.metrics import Metrics

__all__ = ["Agent", "AgentFactory", "AgentT", "Precision"]

AgentT = TypeVar("AgentT", bound="Agent")


class Precision(NamedTuple):
    """What ``autocast=`` asks for: the compute dtype and whether ``torch.autocast`` is entered at all.  ``False`` / ``None``:
    plain fp32; ``True``: fp16 autocast (the reference's default for a bare flag); a dtype or its name: autocast to it.  A
    GradScaler is only needed for fp16 (bf16 has fp32's exponent range)."""

    dtype: torch.dtype
    enabled: bool

    _NAMES = {"float16": torch.float16, "fp16": torch.float16, "half": torch.float16, "bfloat16": torch.bfloat16,
              "bf16": torch.bfloat16, "float32": torch.float32, "fp32": torch.float32}

    @classmethod
    def resolve(cls, autocast: bool | None | torch.dtype | str) -> "Precision":
        if isinstance(autocast, torch.dtype):
            return cls(autocast, True)
        if isinstance(autocast, str):
            name = autocast.removeprefix("torch.")
            if name not in cls._NAMES:
                raise ValueError(f"Unknown autocast dtype '{autocast}' (one of {sorted(cls._NAMES)})")
            return cls(cls._NAMES[name], True)
        return cls(torch.float16, True) if autocast else cls(torch.float32, False)

    @property
    def needs_grad_scaler(self) -> bool:
        return self.enabled and self.dtype is torch.float16


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
    """Decorator of ``act``: the action leaves in the container, on the device and (for floating actions) in the dtype the
    observation came in — a NumPy env gets NumPy actions, a torch env tensors where it keeps its own (agent.py:373-391)."""

    @functools.wraps(act)
    def in_callers_format(self, observation, state=None):
        action: torch.Tensor = act(self, observation, state)
        floating = torch.is_floating_point(action)
        if isinstance(observation, np.ndarray):
            array = action.cpu().numpy()
            return array.astype(observation.dtype) if floating else array
        return action.to(device=observation.device, dtype=observation.dtype if floating else None)

    return in_callers_format


class Agent(ABC):
    Factory = AgentFactory
    MODULES: list[str] = []    # attribute names of nn.Module-like parts (parameters, train / eval, checkpointed)
    STATEFULS: list[str] = []  # attribute names of further checkpointed parts (optimizer, grad scaler)

    def __init__(self, environment_spec: EnvironmentSpec, num_steps_per_update: int, name: str = "Agent",
                 device=None, compile: bool | str = False, autocast: bool | None | torch.dtype | str = False):
        self.environment_spec = environment_spec
        self.observation_dim, self.action_dim = environment_spec.observation_dim, environment_spec.action_dim
        self.has_state = environment_spec.state_dim is not None
        self.state_dim = environment_spec.state_dim if self.has_state else environment_spec.observation_dim
        self.parallelism = environment_spec.num_instances
        self.num_steps_per_update, self.name = num_steps_per_update, name
        self.device = resolve_device(device)
        self.compile = compile
        self.precision = Precision.resolve(autocast)
        self.inference_mode = self.deterministic = False
        self.transition: dict[str, Any] = {}
        self.metrics = Metrics()
        self.iteration = self.step_index = 0

    # ------------------------------------------------------------------ precision
    @property
    def dtype(self) -> torch.dtype:
        return self.precision.dtype

    @property
    def autocast_enabled(self) -> bool:
        return self.precision.enabled

    @property
    def grad_scaler_enabled(self) -> bool:
        return self.precision.needs_grad_scaler

    def autocast(self):
        """Context of every forward pass (``with agent.autocast():``) — a no-op object on the fp32 path."""
        if not self.precision.enabled:
            return nullcontext()
        return torch.autocast(device_type=self.device.type, dtype=self.precision.dtype, enabled=True)

    # ------------------------------------------------------------------ the named parts
    def _parts(self, names: Iterable[str] | None = None) -> Iterator[tuple[str, Any]]:
        """``(name, object)`` of the parts that exist, modules first (``MODULES`` then ``STATEFULS`` unless ``names`` is given)."""
        for name in (self.MODULES + self.STATEFULS if names is None else names):
            part = getattr(self, name, None)
            if part is not None:
                yield name, part

    def named_parameters(self) -> Iterable[tuple[str, torch.nn.Parameter]]:
        for name, module in self._parts(self.MODULES):
            yield from module.named_parameters(prefix=name)

    def parameters(self):
        return (param for _, param in self.named_parameters())

    def setup_module(self, module):
        return module.to(device=self.device)

    def _set_training_mode(self, mode: bool = True):
        for _, module in self._parts(self.MODULES):
            module.train(mode)

    @contextmanager
    def _training_mode(self):
        self._set_training_mode(True)
        try:
            yield
        finally:
            self._set_training_mode(False)

    # ------------------------------------------------------------------ the loop
    @abstractmethod
    def act(self, observation, state=None): ...

    @abstractmethod
    def step(self, next_observation, reward, terminated, truncated, next_state=None, **kwargs) -> bool:
        """One env step collected; True once ``num_steps_per_update`` of them are (never in inference mode) — agent.py:210-213."""
        if self.inference_mode:
            return False
        self.step_index += 1
        return self.step_index >= self.num_steps_per_update

    @abstractmethod
    def update(self) -> dict[str, float]:
        """Close an update: cadence counter back to zero, one more iteration, the recorded metrics as ``{name/metric: value}``
        (one host copy, utils/metrics.py) and an empty metric store."""
        self.step_index = 0
        self.iteration += 1
        # `deferred_summary` (set by a Trainer that pipelines its logging, template/trainer.py): the copy to the host is issued
        # here, the values are formed when the caller asks for them — `StagedSummary.resolve()` — and the caller may enqueue
        # the next rollout in between instead of leaving the device idle behind a blocking read
        staged = self.metrics.staged_summary(self.name)
        self.metrics.clear()
        return staged if getattr(self, "deferred_summary", False) else staged.resolve()

    def set_inference_mode(self, mode: bool = True, deterministic: bool | None = True):
        """Inference: no buffer writes, no updates; ``deterministic`` (None: leave as it is) only ever holds while inferring."""
        self.inference_mode = mode
        if deterministic is not None:
            self.deterministic = mode and deterministic

    def set_iteration(self, iteration: int):
        if iteration < 0:
            raise ValueError("Iteration must be non-negative")
        self.iteration = iteration

    def record(self, metrics: Mapping[str, Any] | None = None, /, **kwargs):
        self.metrics.record(metrics, **kwargs)

    # ------------------------------------------------------------------ tensors
    def to_tensor(self, value: Any) -> torch.Tensor:
        """``value`` on the agent's device, never aliasing the caller's tensor (an env may overwrite its outputs in place)."""
        tensor = torch.as_tensor(value, device=self.device)
        return tensor.clone() if tensor is value else tensor

    def to_nested_tensor(self, value):
        if value is None:
            return None
        if isinstance(value, Mapping):
            return {key: self.to_nested_tensor(item) for key, item in value.items()}
        if isinstance(value, (tuple, list)):
            return tuple(map(self.to_nested_tensor, value))
        return self.to_tensor(value)

    # ------------------------------------------------------------------ checkpoints
    def state_dict(self):
        return {name: part.state_dict() for name, part in self._parts()}

    def load_state_dict(self, state_dict: dict[str, Any]):
        """Restore every part that has an entry; parts without one, entries without a part and parts that reject their entry
        are reported (rank 0) instead of raised — a checkpoint of a differently composed agent still restores what it shares."""
        parts = dict(self._parts())
        absent = [name for name in parts if state_dict.get(name) is None]
        unknown = [key for key in state_dict if key not in parts]
        rejected = {}
        for name, part in parts.items():
            if name in absent:
                continue
            try:
                part.load_state_dict(state_dict[name])
            except (RuntimeError, ValueError) as error:
                rejected[name] = error
        for name in absent:
            self.warn(f"checkpoint has nothing for '{name}': left as it is")
        for name, error in rejected.items():
            self.warn(f"checkpoint entry '{name}' does not fit: {error}")
        if unknown:
            self.warn(f"checkpoint entries without a counterpart: {sorted(unknown)}")

    @classmethod
    def warn(cls, message):
        distributed.print_rank0(f"\033[1;33m{cls.__name__}: {message}\033[0m")
