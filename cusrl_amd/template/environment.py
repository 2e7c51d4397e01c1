"""Environment contract (counterpart of cusrl/template/environment.py:24-379), trimmed to what the rollout and
update loops read.  Vectorised envs return ``[N, ...]`` arrays; ``terminated`` / ``truncated`` are ``[N, 1]`` bools.
"""

from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any

import torch

__all__ = ["Environment", "EnvironmentSpec", "get_done_indices", "update_observation_and_state"]

_SPEC_DEFAULTS: dict[str, Any] = dict(
    action_denormalization=None,
    action_space=None,
    autoreset=False,
    demonstration_sampler=None,
    environment_instance=None,
    final_state_is_missing=False,
    mirror_action=None,
    mirror_observation=None,
    mirror_state=None,
    num_instances=1,
    observation_is_subset_of_state=None,
    observation_stat_groups=(),
    observation_normalization=None,
    observation_normalization_excluded_indices=None,
    observation_space=None,
    reward_dim=1,
    state_dim=None,
    state_stat_groups=(),
    state_normalization=None,
    state_normalization_excluded_indices=None,
    timestep=None,
)


class EnvironmentSpec:
    """Static properties of an environment (same attribute names as the reference spec, ``:24-175``)."""

    def __init__(self, observation_dim: int, action_dim: int, *, device: torch.device | str = "cpu", **properties):
        self.observation_dim = observation_dim
        self.action_dim = action_dim
        self.device = torch.device(device)
        for name, default in _SPEC_DEFAULTS.items():
            setattr(self, name, properties.pop(name, default))
        self.observation_stat_groups = tuple(self.observation_stat_groups)
        self.state_stat_groups = tuple(self.state_stat_groups)
        for name, value in properties.items():  # free-form extras, like the reference's **kwargs
            setattr(self, name, value)

    def get(self, key: str, default=None):
        return self.__dict__.get(key, default)


class Environment(ABC):
    Spec = EnvironmentSpec

    def __init__(self, observation_dim: int, action_dim: int, *, num_instances: int = 1, state_dim: int | None = None,
                 device: torch.device | str = "cpu", **properties):
        self.num_instances = num_instances
        self.observation_dim = observation_dim
        self.action_dim = action_dim
        self.state_dim = state_dim
        self.spec = EnvironmentSpec(
            observation_dim, action_dim, device=device, num_instances=num_instances, state_dim=state_dim,
            environment_instance=self, **properties,
        )

    def close(self):
        pass

    @abstractmethod
    def reset(self, *, indices=None, randomize_episode_progress: bool = False):
        """-> (observation [Ni, Do], state [Ni, Ds] | None, info dict) for the reset instances."""

    @abstractmethod
    def step(self, action):
        """-> (next_observation, next_state | None, reward [N, Dr], terminated [N,1] bool, truncated [N,1] bool, info)."""

    # ---- extension: the capturable protocol (template/graphs.py GraphedRolloutStep)
    capturable: bool = False
    """True promises that ``step`` and ``reset_static`` only enqueue shape-static device work on torch's current stream:
    no host read-back, no Python state that changes from step to step, outputs that are fresh tensors.  ``Trainer``
    then drives such an env without a single host synchronisation per step and, under ``compile=True``, replays a whole
    env step (act -> step -> episode statistics -> post_step hooks -> buffer push -> resets) from ONE hipGraph."""

    generator_free: bool = False
    """True promises, on top of ``capturable``, that ``step`` and ``reset_static`` draw nothing from torch's global generator
    (a simulator with a random stream of its own).  The only consumer of that generator inside a captured rollout is then the
    policy's exploration noise, one draw per env step (cusrl/nn/module/distribution.py:256-262) — and ``GraphedRolloutStep`` issues
    the T draws of a rollout AHEAD of it, on a side stream while the previous update still runs: the same calls in the same
    order, hence the same numbers, off the rollout's serial chain of launches."""

    def reset_static(self, indices: torch.Tensor, count: torch.Tensor):
        """Fixed-shape form of ``reset(indices=...)`` (environment.py:300-317 of the reference takes a host index list):
        ``indices`` is an int64 device vector of env ids of which only the first ``count`` (1-element int32 device
        tensor, read by kernels — never by the host) are finished envs, in ascending order; entries past ``count`` are
        stale but valid env ids whose returned rows are dropped.  Returns ``(observation [len(indices), Do],
        state [len(indices), Ds] | None, info)``; row ``k`` belongs to env ``indices[k]``.  An env with per-instance
        state of its own must only re-initialise the first ``count`` entries (mask by ``arange < count``)."""
        raise NotImplementedError(f"{type(self).__name__} does not implement the capturable reset protocol")

    def get_metrics(self) -> dict[str, float]:
        return {}

    def state_dict(self) -> dict[str, Any]:
        return {}

    def load_state_dict(self, state_dict: dict[str, Any]):
        pass


def get_done_indices(terminated, truncated) -> list[int]:
    """Host list of finished env ids (``:356-362``) — a device->host sync when the env lives on the GPU."""
    done = (terminated | truncated).squeeze(-1)
    nz = done.nonzero()
    if isinstance(nz, tuple):  # numpy
        nz = nz[0]
    return nz.reshape(-1).tolist()


def update_observation_and_state(last_observation, last_state, indices, init_observation, init_state):
    """Splice reset observations into the rollout's current observation (``:365-379``)."""
    if init_observation.shape == last_observation.shape:
        return init_observation, init_state
    last_observation[indices] = init_observation
    if last_state is not None:
        last_state[indices] = init_state
    return last_observation, last_state
