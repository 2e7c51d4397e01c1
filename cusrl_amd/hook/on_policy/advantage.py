"""Advantage post-processing (replaces cusrl/hook/on_policy/advantage.py:13-115).

``AdvantageNormalization``: per-channel ``var_mean`` (unbiased) over all leading dims, optional cross-rank
equal-weight merge, then ``(adv - mean) / sqrt(var + 1e-8)`` in place with a true division.  When it runs right
after the GAE hook on a buffer, the statistics pass is free: the GAE kernel already produced the block partials.
"""

from __future__ import annotations

from collections.abc import Sequence
from typing import Any, Literal

import torch
from torch import Tensor

from cusrl_amd import ops
from cusrl_amd.template.buffer import Buffer
from cusrl_amd.template.hook import Hook
from cusrl_amd.utils import distributed

__all__ = ["AdvantageNormalization", "AdvantageReduction"]


_CHANNEL_REDUCTIONS = {"sum": torch.sum, "mean": torch.mean}


class AdvantageReduction(Hook):
    """Collapse a multi-channel advantage ``[..., C]`` to ``[..., 1]`` inside ``objective``: a (weighted) sum or mean over the
    channels (surface and arithmetic of cusrl/hook/on_policy/advantage.py:13-71: product with the weights first, then the
    reduction; ``weight`` is a mutable attribute).  The weights live in ONE place — :meth:`_set_weight` keeps the tuple the
    schedule sees and the device tensor the objective multiplies with in step, for ``init`` and for ``update_attribute`` alike."""

    def __init__(self, reduction: Literal["sum", "mean"] = "sum", weight: Sequence[float] | None = None):
        if reduction not in _CHANNEL_REDUCTIONS:
            raise ValueError(f"Unsupported reduction '{reduction}'")
        super().__init__(training_only=True)
        self.reduction = reduction
        self.weight: tuple[float, ...] | None = None
        self._weight_tensor: Tensor | None = None
        self._set_weight(weight, on_device=False)
        self.register_mutable("weight")

    def _set_weight(self, weight: Sequence[float] | None, on_device: bool = True):
        self.weight = None if weight is None else tuple(weight)
        self._weight_tensor = self.agent.to_tensor(self.weight) if on_device and self.weight is not None else None

    def init(self):
        self._set_weight(self.weight)

    def objective(self, metadata, batch):
        advantage: Tensor = batch["advantage"]
        weighted = advantage if self._weight_tensor is None else advantage * self._weight_tensor
        batch["advantage"] = _CHANNEL_REDUCTIONS[self.reduction](weighted, -1, keepdim=True)

    def update_attribute(self, name: str, value: Any):
        super().update_attribute(name, value)
        if name == "weight":
            self._set_weight(value)


class AdvantageNormalization(Hook):
    def __init__(self, mini_batch_wise: bool = False, synchronize: bool = True):
        super().__init__(training_only=True)
        self.mini_batch_wise = mini_batch_wise
        self.synchronize = synchronize

    def collective_phases(self):
        return ("objective",) if (self.mini_batch_wise and self.synchronize) else ()

    def pre_update(self, buffer):
        if self.mini_batch_wise:
            return
        derived = buffer.take_derived("advantage") if isinstance(buffer, Buffer) else None
        self.normalize_(buffer["advantage"], derived)

    def objective(self, metadata, batch):
        if self.mini_batch_wise:
            self.normalize_(batch["advantage"])

    @torch.no_grad()
    def normalize_(self, advantage: Tensor, derived=None):
        channels = advantage.shape[-1]
        count = advantage.numel() // channels
        if derived is not None and derived[0] == "stat_partials" and derived[2] == count:
            partials = derived[1]  # emitted by the GAE launch for exactly this tensor
        else:
            partials = ops.col_stats(advantage)
        if not (self.synchronize and distributed.enabled()) and advantage.is_contiguous():
            # nothing sits between the statistics and their use: finalize + normalise in one launch
            ops.normalize_from_partials_(advantage, partials, count, 1e-8)
            return
        var, mean = ops.adv_stats_finalize(partials, count)
        if self.synchronize and advantage.is_cuda and advantage.is_contiguous() and channels <= 256:
            # several ranks: finalize (mean | var in one row) -> all-gather -> merge + normalise in one launch — four launches
            # behind the GAE instead of six (finalize, cat, all-gather, merge, normalise); the same operations in the same order
            # as reduce_mean_var_ (distributed.py:175-183) + the normalisation below
            ops.normalize_from_gathered_(advantage, distributed.gather_stack(ops.packed_mean_var(mean, var)), 1e-8)
            return
        if self.synchronize:
            distributed.reduce_mean_var_(mean, var)
        ops.normalize_(advantage, mean, var, 1e-8)
