"""Running mean / standard deviation of a data stream (counterpart of cusrl/nn/layer/rms.py:14-246 and
cusrl/nn/utils/normalization.py:15-93), device-resident on MI355X.

Same buffers (``mean``, ``var``, ``std``), same merge formula (Chan et al., weights ``count : batch_count``),
same group / excluded-channel handling and the same distributed merge — but the sample ``count`` lives on the
device (fp64 scalar) so that a masked update (``observation[done]`` in the reference, a boolean-mask select that
synchronises with the host twice per env step) needs no host round trip: masked column statistics, merge and
normalise are three HIP launches (``cusrl_masked_col_stats``, ``cusrl_rms_merge``, ``cusrl_rms_normalize``).
On CPU tensors the same arithmetic runs as plain torch ops (host-side module state, not the hot path).
"""

from __future__ import annotations

from collections.abc import Iterable
from typing import Any

import torch
from torch import Tensor, nn

from cusrl_amd.utils import distributed

__all__ = ["RunningMeanStd", "mean_var_count", "merge_mean_var_"]


def mean_var_count(input: Tensor, mask: Tensor | None = None) -> tuple[Tensor, Tensor, Tensor]:
    """Per-channel mean, population variance and (tensor) count of ``input [..., C]`` over the rows selected by
    ``mask``; an empty selection gives (0, 1, 0)."""
    if input.ndim < 2:
        raise ValueError("Input tensor must be at least 2-dimensional")
    input = input.flatten(0, -2)
    if input.is_cuda:
        from cusrl_amd import ops

        return ops.masked_col_stats(input.float(), None if mask is None else mask.reshape(-1))
    from cusrl_amd.utils.misc import host_form

    host_form("RunningMeanStd (mean_var_count)")  # test processes without a GPU only
    if mask is not None:
        input = input[mask.reshape(-1).bool()]
    count = torch.tensor([float(input.size(0))], dtype=torch.float64)
    if input.size(0) == 0:
        return input.new_zeros(input.size(1)), input.new_ones(input.size(1)), count
    var, mean = torch.var_mean(input, dim=0, correction=0)
    return mean, var, count


def merge_mean_var_(old_mean: Tensor, old_var: Tensor, w_old, new_mean: Tensor, new_var: Tensor, w_new):
    """Weighted in-place merge with host weights (normalization.py:80-93); used by the distributed bookkeeping."""
    w_sum = w_old + w_new
    if w_sum <= 0:
        raise ValueError(f"Weight sum must be positive; got {w_sum}")
    w_old, w_new = w_old / w_sum, w_new / w_sum
    delta = new_mean - old_mean
    old_mean.add_(delta * w_new)
    old_var.add_((new_var - old_var) * w_new + delta.square() * (w_old * w_new))


class RunningMeanStd(nn.Module):
    def __init__(self, num_channels: int, *, groups: Iterable = (), excluded_indices=None, clamp: float | None = 10.0,
                 max_count: int | None = None, epsilon: float = 1e-8):
        if clamp is not None and clamp <= 0:
            raise ValueError("'clamp' must be None or a positive value")
        if max_count is not None and max_count <= 0:
            raise ValueError("'max_count' must be None or a positive value")
        super().__init__()
        self.groups = tuple(groups)
        self.excluded_indices = excluded_indices
        self.clamp, self.max_count, self.epsilon = clamp, max_count, epsilon
        usage = torch.zeros(num_channels, dtype=torch.int64)
        for indices in self.groups:
            usage[indices,] += 1
        if torch.any(usage > 1):
            raise ValueError("Indices in 'groups' must not overlap")
        if excluded_indices is not None:
            excluded = torch.zeros(num_channels, dtype=torch.bool)
            excluded[excluded_indices,] = True
            if torch.any(usage[excluded] > 0):
                raise ValueError("'excluded_indices' must not overlap with 'groups'")
        self.register_buffer("mean", torch.zeros(num_channels))
        self.register_buffer("var", torch.ones(num_channels))
        self.register_buffer("std", torch.ones(num_channels))
        self.register_buffer("_count", torch.zeros(1, dtype=torch.float64), persistent=False)
        self._is_synchronized = True
        self._synchronized_state: tuple[Tensor, Tensor, Tensor] | None = None

    # ---- count: device-resident, readable as an int (one host read)
    @property
    def count(self) -> int:
        return int(round(self._count.item()))

    @count.setter
    def count(self, value: int):
        self._count.fill_(float(value))

    def clear(self):
        self.mean.fill_(0.0)
        self.var.fill_(1.0)
        self.std.fill_(1.0)
        self._count.zero_()
        self._is_synchronized = False
        self._synchronized_state = None

    # ---- updates
    def update(self, input: Tensor, *, mask: Tensor | None = None, synchronize: bool = True):
        self.update_from_stats(*mean_var_count(input, mask), synchronize=synchronize)

    @torch.no_grad()
    def update_from_stats(self, batch_mean: Tensor, batch_var: Tensor, batch_count, *, synchronize: bool = True):
        if not isinstance(batch_count, Tensor):
            batch_count = torch.tensor([float(batch_count)], dtype=torch.float64, device=self.mean.device)
        if synchronize and distributed.enabled():
            self.synchronize()
            batch_mean, batch_var, batch_count = _synchronize_mean_var_count(batch_mean, batch_var, batch_count)
        self._process_mean_var(batch_mean, batch_var)
        capped = self.max_count if synchronize else None
        if self.mean.is_cuda:
            from cusrl_amd import ops

            ops.rms_merge_(self.mean, self.var, self.std, self._count, batch_mean, batch_var, batch_count, self.epsilon, capped)
        else:
            n = float(batch_count.item())
            if n == 0:
                return
            merge_mean_var_(self.mean, self.var, float(self._count.item()), batch_mean, batch_var, n)
            self.std.copy_(torch.sqrt(self.var + self.epsilon))
            total = float(self._count.item()) + n
            self._count.fill_(min(total, capped) if capped is not None else total)
        self._is_synchronized = synchronize
        if synchronize and distributed.enabled():
            self._synchronized_state = (self.mean.clone(), self.var.clone(), self._count.clone())

    def synchronize(self):
        """Merge statistics accumulated locally since the last synchronisation across ranks (rms.py:169-196)."""
        if self._is_synchronized or not distributed.enabled():
            return
        if self._synchronized_state is None:
            total_mean, total_var, total_count = _synchronize_mean_var_count(self.mean, self.var, self._count)
        else:
            sync_mean, sync_var, sync_count = self._synchronized_state
            local, base = float(self._count.item()), float(sync_count.item())
            merge_mean_var_(self.mean, self.var, local, sync_mean, sync_var, -base)  # what this rank added since
            patch = _synchronize_mean_var_count(self.mean, self.var, self._count - sync_count)
            merge_mean_var_(sync_mean, sync_var, base, patch[0], patch[1], float(patch[2].item()))
            total_mean, total_var, total_count = sync_mean, sync_var, sync_count + patch[2]
        self.mean.copy_(total_mean)
        self.var.copy_(total_var)
        self.std.copy_(torch.sqrt(total_var + self.epsilon))
        self._count.copy_(total_count)
        if self.max_count is not None:
            self._count.clamp_(max=float(self.max_count))
        self._is_synchronized = True
        self._synchronized_state = (self.mean.clone(), self.var.clone(), self._count.clone())

    # ---- normalisation
    def forward(self, input: Tensor) -> Tensor:
        return self.normalize(input)

    def normalize(self, input: Tensor) -> Tensor:
        if input.is_cuda and input.dtype == torch.float32 and not input.requires_grad:
            from cusrl_amd import ops

            return ops.rms_normalize(input, self.mean, self.std, self.clamp)
        # differentiable inputs (autograd has to see the ops), other dtypes (autocast), CPU module state
        output = (input - self.mean) / self.std
        if self.clamp is not None:
            output = output.clamp(-self.clamp, self.clamp)
        return output.type_as(input)

    def normalize_(self, input: Tensor) -> Tensor:
        input.sub_(self.mean).div_(self.std)
        return input.clamp_(-self.clamp, self.clamp) if self.clamp is not None else input

    def unnormalize(self, input: Tensor) -> Tensor:
        return (input * self.std + self.mean).type_as(input)

    def _process_mean_var(self, batch_mean: Tensor, batch_var: Tensor):
        """Excluded channels keep (0, 1); grouped channels share pooled statistics (rms.py:221-231)."""
        if self.excluded_indices is not None:
            batch_mean[self.excluded_indices,] = 0.0
            batch_var[self.excluded_indices,] = 1.0
        for indices in self.groups:
            group_mean = batch_mean[indices,].mean()
            group_var = batch_var[indices,].mean() - group_mean.square() + batch_mean[indices,].square().mean()
            batch_mean[indices,] = group_mean
            batch_var[indices,] = group_var

    # ---- checkpoint: the count travels as extra state, like the reference
    def get_extra_state(self) -> Any:
        return torch.tensor(self.count, dtype=torch.int64)

    def set_extra_state(self, state: Any):
        count = int(state.item() if isinstance(state, Tensor) else state)
        if count < 0:
            raise ValueError("'count' must be non-negative")
        self.count = count
        self._is_synchronized = True
        self._synchronized_state = (self.mean.clone(), self.var.clone(), self._count.clone())


def _synchronize_mean_var_count(mean: Tensor, var: Tensor, count: Tensor) -> tuple[Tensor, Tensor, Tensor]:
    """Count-weighted merge of every rank's (mean, var, count) from ONE all-gather (normalization.py:53-77)."""
    if not distributed.enabled():
        return mean, var, count
    packed = torch.cat((mean.double(), var.double(), count.double().reshape(1)), dim=0)
    gathered = distributed.gather_stack(packed)  # [W, 2C + 1]
    dim = mean.size(0)
    means, vars_, counts = gathered[:, :dim], gathered[:, dim : 2 * dim], gathered[:, [2 * dim]]
    total = counts.sum()
    weights = counts / (total + 1e-8)
    total_mean = (means * weights).sum(dim=0)
    total_var = ((vars_ + (means - total_mean).square()) * weights).sum(dim=0)
    empty = total <= 0
    total_mean = torch.where(empty, mean.double(), total_mean).to(mean.dtype)
    total_var = torch.where(empty, var.double(), total_var).to(var.dtype)
    return total_mean, total_var, total.reshape(1)
