"""Plugin lifecycle (counterpart of cusrl/template/hook.py:19-485).

A ``Hook`` sees every phase of the agent: ``pre_init / init / post_init``, ``pre_act / post_act / post_step /
should_update``, ``pre_update(buffer)``, ``pre_objective / objective / pre_optim / post_optim / post_objective`` per
minibatch, ``post_update`` and ``apply_schedule``.  ``HookComposite`` fans calls out in list order, skipping
inactive hooks and training-only hooks while the agent is in inference mode.

MI355X-specific part: when the objective-producing hooks are exactly the stock PPO set, ``HookComposite``
arms a :class:`~cusrl_amd.hook.on_policy.fused.FusedPpoObjective` so the four of them contribute to ONE fused
HIP kernel (loss + gradients) instead of ~40 elementwise/reduction launches; any other composition runs hook by
hook with identical results.
"""

from __future__ import annotations

import itertools
from collections.abc import Iterable, Iterator, Mapping
from typing import Any, Generic

import torch
from torch import nn

from cusrl_amd.template.agent import AgentT
from cusrl_amd.utils import distributed
from cusrl_amd.utils.misc import MISSING, camel_to_snake

__all__ = ["Hook", "HookComposite", "Objectives"]


class Roots(list):
    """The summands of a step's loss as separate autograd roots.  ``branch`` = ``(root, stream)``: that summand (also an
    element of the list) was evaluated on another stream and reaches only the critic's parameters — ``ActorCritic._backward``
    may differentiate it there, concurrently with the others."""

    branch: tuple | None = None


class Objectives(dict):
    """Loss terms by name; ``total`` (optional) is a pre-summed, differentiable scalar covering ``fused_keys``;
    ``branch_root`` (optional) = ``(value term, stream)`` when that fused term was evaluated by its own launch on the critic's
    stream (``total`` then covers the other fused keys only)."""

    total: torch.Tensor | None = None
    fused_keys: tuple[str, ...] = ()
    branch_root: tuple | None = None

    def loss(self) -> torch.Tensor:
        """``sum(objectives.values())`` as actor_critic.py:309 does, with fused terms taken from ``total``."""
        if self.total is None:
            return sum(self.values())
        loss = self.total
        if self.branch_root is not None:
            root, stream = self.branch_root
            torch.cuda.current_stream().wait_stream(stream)
            loss = root + loss  # (value + surrogate) + entropy: the hooks' insertion order
        for key, value in self.items():
            if key not in self.fused_keys:
                loss = loss + value
        return loss

    def terms(self) -> Roots:
        """The summands of :meth:`loss` as separate roots: differentiating them together with unit gradients gives the
        gradient of the sum without launching the additions (one tiny kernel per auxiliary term and minibatch step)."""
        if self.total is None:
            return Roots(value for value in self.values() if value is not None)
        roots = Roots([self.total] + [value for key, value in self.items() if key not in self.fused_keys and value is not None])
        if self.branch_root is not None:
            roots.append(self.branch_root[0])
            roots.branch = self.branch_root
        return roots


class Hook(Generic[AgentT]):
    agent: AgentT
    # Extension: the hook draws from torch's global generator between pre_objective and post_objective (e.g. AMP samples
    # a discriminator batch).  The sampler then keeps every permutation draw exactly where the reference has it.
    objective_draws_random: bool = False
    # Extension: the hook draws from torch's generator inside an ENV STEP (``pre_act`` / ``post_act`` / ``post_step``) — then the
    # exploration noise of a rollout cannot be drawn ahead of it (template/graphs.py GraphedRolloutStep._noise_plan: the draws would
    # change places with the hook's in the generator's stream).
    step_draws_random: bool = False
    # Extension: this hook's post_step / should_update only enqueue shape-static device work (no Python state that changes
    # per step, no host read-back), so a whole env step may be replayed from a hipGraph (template/graphs.py
    # GraphedRolloutStep).  Stock hooks qualify; a user-defined hook that overrides either method sets this to opt in.
    rollout_capture_safe: bool = False

    @property
    def post_step_device_free(self) -> bool:
        """Extension: this hook's ``post_step`` neither reads nor writes device memory of the transition (nothing it does
        sits between the step epilogue and the buffer append on the device) — true for every hook that does not override
        ``post_step``; a hook whose override is host-only says so by overriding this property.  When every active hook is
        device-free the epilogue and the append of a captured env step are ONE launch (``cusrl_step_epilogue_push``)."""
        return type(self).post_step is Hook.post_step

    def __init__(self, training_only: bool = False):
        self._modules: dict[str, nn.Module | None] = {}
        self._statefuls: dict[str, Any] = {}
        self._mutable: set[str] = set()
        self._name = camel_to_snake(type(self).__name__)
        self._active = True
        self._training_only = training_only

    # ---- identity / activation
    @property
    def name(self) -> str:
        return self._name

    @property
    def active(self) -> bool:
        return self._active

    @property
    def training_only(self) -> bool:
        return self._training_only

    def name_(self, name: str):
        self._name = name
        return self

    def active_(self, active: bool):
        self._active = active
        return self

    # ---- registries (hook.py:74-141)
    def register_module(self, name: str, module: nn.Module | None):
        if name in self._statefuls:
            raise RuntimeError(f"Cannot register module '{name}': a stateful with the same name already exists")
        if module is not None:
            module = self.agent.setup_module(module)
        setattr(self, name, module)
        self._modules[name] = module

    def register_stateful(self, name: str, value: Any):
        if name in self._modules:
            raise RuntimeError(f"Cannot register stateful '{name}': a module with the same name already exists")
        setattr(self, name, value)
        self._statefuls[name] = value

    def register_mutable(self, name: str, value: Any = MISSING):
        if value is not MISSING:
            setattr(self, name, value)
        self._mutable.add(name)

    def update_attribute(self, name: str, value: Any):
        if name not in self._mutable:
            raise ValueError(f"Attribute '{name}' is not mutable on hook '{self.name}'")
        setattr(self, name, value)

    # ---- parameters / state
    def named_parameters(self, prefix: str = "") -> Iterator[tuple[str, nn.Parameter]]:
        lead = f"{prefix}." if prefix else ""
        for name, module in self._modules.items():
            if module is not None:
                yield from module.named_parameters(prefix=lead + name)

    def parameters(self):
        for _, param in self.named_parameters():
            yield param

    def _stateful_items(self):
        return itertools.chain(self._modules.items(), self._statefuls.items())

    def state_dict(self):
        return {name: part.state_dict() for name, part in self._stateful_items() if part is not None}

    def load_state_dict(self, state_dict: Mapping[str, Any]):
        unused = set(state_dict)
        for name, part in self._stateful_items():
            if part is None:
                continue
            if name not in state_dict:
                self.warn(f"No state_dict entry was found for '{name}'.")
                continue
            unused.discard(name)
            try:
                part.load_state_dict(state_dict[name])
            except (RuntimeError, ValueError) as error:
                self.warn(f"State dict for '{name}' is incompatible: {error}")
        if unused:
            self.warn(f"Unused state_dict keys: {unused}.")

    def compile(self, **kwargs):
        for module in self._modules.values():
            if module is not None and hasattr(module, "compile"):
                module.compile(**kwargs)

    def train(self, mode: bool = True):
        for module in self._modules.values():
            if module is not None and hasattr(module, "train"):
                module.train(mode)

    def eval(self):
        self.train(False)

    # ---- lifecycle (all no-ops here)
    def pre_init(self, agent: AgentT):
        self.agent = agent

    def init(self): ...

    def post_init(self): ...

    def pre_act(self, transition): ...

    def post_act(self, transition): ...

    def post_step(self, transition): ...

    def should_update(self, transition) -> bool:
        return True

    def pre_update(self, buffer): ...

    def pre_objective(self, metadata, batch): ...

    def objective(self, metadata, batch) -> dict[str, torch.Tensor] | None:
        return None

    def pre_optim(self, optimizer): ...

    def post_optim(self): ...

    def post_objective(self, metadata, batch): ...

    def post_update(self): ...

    def apply_schedule(self, iteration: int): ...

    def collective_phases(self) -> tuple[str, ...]:
        """Extension: phases among ``"act"`` (pre_act .. post_act) and ``"objective"`` (pre_objective .. post_objective)
        in which this hook calls a cross-rank collective when training is distributed.  ``compile=True`` keeps such
        phases out of hipGraph capture (template/graphs.py)."""
        return ()

    def eager_phases(self) -> tuple[str, ...]:
        """Extension: phases (same names, plus ``"step"`` = post_step inside a captured env step) this hook must run
        outside hipGraph capture whatever the number of ranks — typically because it reads a device value back to the
        host to branch on it."""
        return ()

    def on_replay(self, phase: str):
        """Extension: called instead of the hook's ``pre_act`` / ``post_act`` (``phase == "act"``) or ``post_step``
        (``"step"``) when that phase was replayed from a hipGraph — the device work happened, the Python body did not.
        A hook whose body also changes HOST state (a flag, a counter) repeats that part here."""

    def pre_export(self, graph): ...

    def post_export(self, graph): ...

    @classmethod
    def warn(cls, message):
        distributed.print_rank0(f"\033[1;31m{cls.__name__}: {message}\033[0m")


def _fan_out(method: str):
    def call(self, *args):
        for hook in self.active_hooks():
            getattr(hook, method)(*args)

    call.__name__ = method
    return call


class HookComposite(Hook):
    """Runs a fixed, uniquely named sequence of hooks (hook.py:364-485)."""

    def __init__(self, hooks: Iterable[Hook]):
        super().__init__()
        self._hooks = tuple(hooks)
        self._named_hooks: dict[str, Hook] = {}
        for hook in self._hooks:
            if not isinstance(hook, Hook):
                raise TypeError(f"Expected a Hook instance, but got '{type(hook).__name__}'")
            if hook.name in self._named_hooks:
                raise RuntimeError(f"Hook '{hook.name}' already exists")
            self._named_hooks[hook.name] = hook
        self._statefuls.update(self._named_hooks)
        self._fusion_checked = False
        self._fusion = None

    def __getitem__(self, name: str) -> Hook:
        if "." in name:
            head, rest = name.split(".", 1)
            return self._named_hooks[head][rest]
        return self._named_hooks[name]

    def __iter__(self) -> Iterator[Hook]:
        yield from self._hooks

    def named_parameters(self, prefix: str = ""):
        lead = prefix if not prefix or prefix.endswith(".") else prefix + "."
        for name, hook in self._named_hooks.items():
            yield from hook.named_parameters(prefix=lead + name)

    def active_hooks(self) -> Iterator[Hook]:
        # called for every lifecycle event of every env step: plain attribute reads instead of the two properties
        if self.agent.inference_mode:
            return iter([hook for hook in self._hooks if hook._active and not hook._training_only])
        return iter([hook for hook in self._hooks if hook._active])

    def compile(self, **kwargs):
        for hook in self:
            hook.compile(**kwargs)

    def train(self, mode=True):
        for hook in self:
            hook.train(mode)

    def pre_init(self, agent):
        super().pre_init(agent)
        for hook in self.active_hooks():
            hook.pre_init(agent)

    init = _fan_out("init")
    post_init = _fan_out("post_init")
    pre_act = _fan_out("pre_act")
    post_act = _fan_out("post_act")
    post_step = _fan_out("post_step")
    pre_update = _fan_out("pre_update")
    pre_objective = _fan_out("pre_objective")
    pre_optim = _fan_out("pre_optim")
    post_optim = _fan_out("post_optim")
    post_objective = _fan_out("post_objective")
    post_update = _fan_out("post_update")
    apply_schedule = _fan_out("apply_schedule")
    on_replay = _fan_out("on_replay")

    def should_update(self, transition) -> bool:
        return all(hook.should_update(transition) for hook in self.active_hooks())

    def pre_export(self, graph):
        for hook in self:
            hook.pre_export(graph)

    def post_export(self, graph):
        for hook in self:
            hook.post_export(graph)

    # ---- the objective, optionally through the fused PPO kernel
    def objective(self, metadata, batch) -> Objectives | None:
        from cusrl_amd.hook.on_policy.fused import FusedPpoObjective  # hooks import this module: keep it lazy

        context = FusedPpoObjective.arm(self, batch)
        objectives = Objectives()
        try:
            for hook in self.active_hooks():
                if (terms := hook.objective(metadata, batch)) is not None:
                    objectives.update(terms)
            if context is not None:
                context.resolve(objectives, batch)
        finally:
            FusedPpoObjective.disarm(self)
        return objectives or None
