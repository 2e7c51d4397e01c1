"""Count-weighted running means of recorded tensors (counterpart of cusrl/utils/metrics.py:12-96).

The reference launches one ``mean()`` kernel per metric per minibatch and reads each metric back with
``.item()``.  Here device values are kept on the device and all metrics are fetched with ONE host copy in
``summary()``; the weighted-mean arithmetic is unchanged.
"""

from __future__ import annotations

import itertools
from collections.abc import Mapping
from typing import Any

import torch

__all__ = ["Metrics"]


class Metric:
    __slots__ = ("mean", "count")

    def __init__(self):
        self.mean: torch.Tensor = torch.tensor([])
        self.count: int = 0

    @torch.no_grad()
    def update(self, mean: torch.Tensor, count: int):
        if count == 0:
            return
        if self.count == 0:
            self.mean, self.count = mean.clone(), count
            return
        total = self.count + count
        self.mean.mul_(self.count / total).add_(mean.to(self.mean.device) * (count / total))
        self.count = total


class Metrics:
    def __init__(self):
        self._data: dict[str, Metric] = {}

    def clear(self):
        self._data.clear()

    def __getitem__(self, name: str) -> Metric:
        return self._data[name]

    def __iter__(self):
        return iter(self._data)

    def __len__(self):
        return len(self._data)

    def items(self):
        return self._data.items()

    def keys(self):
        return self._data.keys()

    def values(self):
        return self._data.values()

    def get(self, name: str, default=None):
        return self._data.get(name, default)

    @torch.no_grad()
    def record(self, metrics: Mapping[str, Any] | None = None, /, **kwargs: Any):
        for name, value in itertools.chain((metrics or {}).items(), kwargs.items()):
            if value is None:
                continue
            try:
                value = torch.as_tensor(value, dtype=torch.float32)
            except Exception as error:
                raise ValueError(f"Failed to update metric '{name}'") from error
            if (numel := value.numel()) == 0:
                continue
            self._data.setdefault(name, Metric()).update(value.mean(), numel)

    def summary(self, prefix: str = "") -> dict[str, float]:
        if prefix and not prefix.endswith("/"):
            prefix += "/"
        if not self._data:
            return {}
        names = list(self._data)
        means = [self._data[n].mean.reshape(()) for n in names]
        by_device: dict[torch.device, list[int]] = {}
        for i, m in enumerate(means):
            by_device.setdefault(m.device, []).append(i)
        values = [0.0] * len(names)
        for idx in by_device.values():  # one host copy per device instead of one .item() per metric
            for i, v in zip(idx, torch.stack([means[i] for i in idx]).tolist()):
                values[i] = v
        return {f"{prefix}{n}": v for n, v in zip(names, values)}
