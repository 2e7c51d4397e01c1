"""GAE(lambda) as one fused HIP launch (replaces cusrl/hook/on_policy/gae.py:8-110).

The reference's ``_generalized_advantage_estimation`` issues ~4 + 3(T-1) tiny torch kernels (delta, then a Python
loop of in-place mul/add per step) and a second full scan when ``lamda_value`` is set.  Here ``cusrl_gae`` computes
delta, both scans, ``return = value + advantage`` and the per-channel {sum, sumsq} of the advantage in ONE pass
(21 B per sample), bit-exact with the reference (separate fp32 multiply and add, same association).
"""

from __future__ import annotations

import torch

from cusrl_amd import ops
from cusrl_amd.template.buffer import Buffer
from cusrl_amd.template.hook import Hook

__all__ = ["GeneralizedAdvantageEstimation"]


def _generalized_advantage_estimation(reward, done, value, next_value, gamma: float, lamda: float) -> torch.Tensor:
    """Functional form with the reference's signature (gae.py:8-20); returns a new advantage tensor."""
    advantage, _, _ = ops.gae(reward, value, next_value, done, gamma, lamda, None, with_stats=False)
    return advantage


class GeneralizedAdvantageEstimation(Hook):
    """Writes ``advantage`` and ``return`` into the buffer before the update (or into each temporal minibatch when
    ``recompute``).  ``lamda_value`` gives the value target its own lambda (DNA, gae.py:33-36)."""

    def __init__(self, gamma: float = 0.99, lamda: float = 0.95, lamda_value: float | None = None, recompute: bool = False):
        if gamma < 0 or gamma >= 1:
            raise ValueError(f"'gamma' must be in [0, 1); got {gamma}")
        if lamda < 0 or lamda > 1:
            raise ValueError(f"'lamda' must be in [0, 1]; got {lamda}")
        if lamda_value is not None and (lamda_value < 0 or lamda_value > 1):
            raise ValueError(f"'lamda_value' must be in [0, 1]; got {lamda_value}")
        super().__init__(training_only=True)
        self.recompute = recompute
        self.gamma: float = gamma
        self.lamda: float = lamda
        self.lamda_value: float | None = lamda_value
        for name in ("gamma", "lamda", "lamda_value"):
            self.register_mutable(name)

    def pre_update(self, buffer):
        if not self.recompute:
            self._compute_advantage_and_return(buffer)

    def objective(self, metadata, batch):
        if self.recompute:
            self._compute_advantage_and_return(batch)

    @torch.no_grad()
    def _compute_advantage_and_return(self, data):
        reward, value, next_value, done = data["reward"], data["value"], data["next_value"], data["done"]
        if isinstance(data, Buffer):
            # results land directly in buffer-owned leaves; the statistics ride along for AdvantageNormalization
            advantage, ret = data.field("advantage", value), data.field("return", value)
            _, _, partials = ops.gae(reward, value, next_value, done, self.gamma, self.lamda, self.lamda_value, advantage, ret)
            data.set_derived("advantage", ("stat_partials", partials, advantage.numel() // advantage.shape[-1]))
        else:
            advantage, ret, _ = ops.gae(reward, value, next_value, done, self.gamma, self.lamda, self.lamda_value, with_stats=False)
            data["advantage"], data["return"] = advantage, ret
