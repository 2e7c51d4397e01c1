"""Synthetic vectorised environments.

``SyntheticEnvironment`` is the benchmark task of BASELINE.json / SURVEY.md §8d: observations and rewards are
i.i.d. N(0, 1), ``terminated ~ Bernoulli(0.01)``, ``truncated ~ Bernoulli(0.005)``, no privileged state, manual
reset returning fresh N(0, 1) rows — everything generated on the env's device from torch's global generator.
``DummyTorchEnvironment`` mirrors cusrl/testing/environment.py:39-63 (10 % / 10 % termination / truncation).
"""

from __future__ import annotations

import os

import torch

from cusrl_amd.template.environment import Environment
from cusrl_amd.utils.config import device as resolve_device

__all__ = ["DummyTorchEnvironment", "SyntheticEnvironment"]


class SyntheticEnvironment(Environment):
    def __init__(self, num_instances: int = 4096, observation_dim: int = 48, action_dim: int = 12, *,
                 state_dim: int | None = None, reward_dim: int = 1, terminate_prob: float = 0.01,
                 truncate_prob: float = 0.005, device=None, autoreset: bool = False, capturable: bool = True,
                 fused: bool | None = None, seed: int | None = None, **properties):
        device = resolve_device(device)
        super().__init__(observation_dim, action_dim, num_instances=num_instances, state_dim=state_dim,
                         reward_dim=reward_dim, device=device, autoreset=autoreset, **properties)
        self.device = device
        self.terminate_prob, self.truncate_prob = terminate_prob, truncate_prob
        # every step / reset is shape-static device work from torch's generator: the trainer may drive it without
        # host synchronisation and replay whole env steps from hipGraphs (template/environment.py `capturable`)
        self.capturable = bool(capturable) and self.device.type == "cuda"
        self._flag_probs = torch.tensor([terminate_prob, truncate_prob], dtype=torch.float32, device=self.device).view(2, 1, 1)
        # fused (default on a GPU, CUSRL_FUSED_ENV=0 / fused=False for the torch-generator form): a whole step — observation,
        # reward, both flags and the rows for the resets — from ONE HIP launch (`cusrl_synthetic_env_step`: Philox keyed on a
        # seed drawn from torch's generator here and a device-resident step counter) instead of five generator launches; the
        # same distributions, another random stream.  No privileged state in this form.
        if fused is None:
            fused = os.environ.get("CUSRL_FUSED_ENV", "1") != "0"
        self.fused = bool(fused) and self.device.type == "cuda" and state_dim is None
        # (the fused step and the reset rows it leaves behind come from the env's own Philox stream: nothing is drawn from torch's
        # generator while the trainer drives the env — template/environment.py `generator_free`)
        self.generator_free = self.fused and self.capturable
        if self.fused:
            # follows set_global_seed (seed + rank) without consuming any generator; `seed=` gives an instance its own stream
            base = torch.initial_seed() if seed is None else int(seed)
            self._seed = (base * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019) & 0xFFFFFFFFFFFFFFFF
            self._counter = torch.zeros(2, dtype=torch.int64, device=self.device)
            self._reset_rows: torch.Tensor | None = None

    def _randn(self, rows: int, cols: int | None):
        return None if cols is None else torch.randn(rows, cols, device=self.device)

    def reset(self, *, indices=None, randomize_episode_progress: bool = False):
        rows = self.num_instances if indices is None else len(indices)
        return self._randn(rows, self.observation_dim), self._randn(rows, self.state_dim), {}

    def reset_static(self, indices, count):
        # the trainer always asks for one row per env slot (`indices` has `num_instances` entries, `count` of them meaningful): those
        # rows come out of the step's own launch.  Any other request (a caller resetting a subset by hand) is served from torch's
        # generator — the same N(0, 1) rows, another stream; the fused stream stays what the step counter says it is.
        if self.fused and self._reset_rows is not None and indices.numel() == self.num_instances:
            return self._reset_rows, None, {}
        rows = indices.numel()  # one fresh row per index slot; the trainer's splice only takes the first `count`
        return self._randn(rows, self.observation_dim), self._randn(rows, self.state_dim), {}

    def step(self, action):
        assert isinstance(action, torch.Tensor) and action.shape == (self.num_instances, self.action_dim)
        n = self.num_instances
        if self.fused:
            from cusrl_amd import ops

            next_observation, reward, terminated, truncated, self._reset_rows = ops.synthetic_env_step(
                self._seed, self._counter, n, self.observation_dim, self.spec.reward_dim, self.terminate_prob, self.truncate_prob)
            return next_observation, None, reward, terminated, truncated, {}
        # both flag vectors from one draw and ONE comparison against the [2, 1, 1] probabilities (the same float32
        # thresholds a Python scalar would be rounded to): [0] = terminated, [1] = truncated, each a contiguous [n, 1]
        flags = torch.rand(2, n, 1, device=self.device) < self._flag_probs
        # with autoreset=True a finished instance restarts from a fresh N(0, 1) observation — which the i.i.d.
        # next observation below already is, so no masked overwrite is needed
        return (
            self._randn(n, self.observation_dim),
            self._randn(n, self.state_dim),
            self._randn(n, self.spec.reward_dim),
            flags[0],
            flags[1],
            {},
        )


class DummyTorchEnvironment(SyntheticEnvironment):
    def __init__(self, *args, **kwargs):
        kwargs.setdefault("terminate_prob", 0.1)
        kwargs.setdefault("truncate_prob", 0.1)
        super().__init__(*args, **kwargs)
