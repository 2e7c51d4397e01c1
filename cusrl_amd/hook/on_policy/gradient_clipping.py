"""Gradient-norm clipping before the optimizer step (counterpart of cusrl/hook/on_policy/gradient_clipping.py:8-83).

Parameters fall into the longest matching name prefix of ``groups`` (default group otherwise); each group is clipped
to its own max norm and its pre-clip norm is recorded as ``grad_norm/<prefix|default>``.  When every parameter
is in the default group and the agent keeps its gradients in one flat buffer, the norm is a single reduction over
that buffer (``cusrl_clip_grad_norm``: norm and scale in two launches) instead of a per-tensor foreach chain.
"""

from __future__ import annotations

from torch import nn

from cusrl_amd import ops
from cusrl_amd.template.hook import Hook

__all__ = ["GradientClipping"]


class GradientClipping(Hook):
    def __init__(self, max_grad_norm: float | None = 1.0, groups: dict[str, float | None] | None = None, **kwargs: float | None):
        super().__init__(training_only=True)
        if max_grad_norm is not None and max_grad_norm < 0:
            raise ValueError("'max_grad_norm' must be non-negative")
        self.max_grad_norm = max_grad_norm
        merged = {**(groups or {}), **kwargs}
        for prefix, limit in merged.items():
            if not prefix:
                raise ValueError("Empty prefixes are not allowed; use 'max_grad_norm' for the default group")
            if limit is not None and limit < 0:
                raise ValueError(f"'max_grad_norm' for prefix '{prefix}' must be non-negative")
        self.groups = dict(sorted(merged.items(), key=lambda item: len(item[0]), reverse=True))

    def _match_prefix(self, name: str) -> str:
        for prefix in self.groups:
            if name == prefix or name.startswith(prefix + "."):
                return prefix
        return ""

    def pre_optim(self, optimizer):
        flat = getattr(self.agent, "flat_gradients", None)
        if not self.groups and flat is not None and flat.intact():
            if self.max_grad_norm is not None:
                flat_optimizer = getattr(self.agent, "flat_optimizer", None)
                if flat_optimizer is not None and flat_optimizer.optimizer is optimizer:
                    # the flat Adam step applies the coefficient while it streams the gradient: one launch here
                    total = flat_optimizer.defer_clip(self.max_grad_norm)
                else:
                    total = ops.clip_grad_norm_(flat.buffer, self.max_grad_norm)  # two launches instead of six
                self.agent.record(**{"grad_norm/default": total})
            return
        buckets: dict[str, list] = {"": [], **{prefix: [] for prefix in self.groups}}
        for group in optimizer.param_groups:
            params = group["params"]
            for param, name in zip(params, group.get("param_names", [""] * len(params)), strict=True):
                buckets[self._match_prefix(name)].append(param)
        for prefix, params in buckets.items():
            limit = self.groups.get(prefix, self.max_grad_norm)
            if params and limit is not None:
                self.agent.record(**{f"grad_norm/{prefix or 'default'}": nn.utils.clip_grad_norm_(params, limit)})
