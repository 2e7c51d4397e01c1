"""Allocator hygiene after an update (counterpart of cusrl/hook/control/empty_cuda_cache.py:8-13)."""

from __future__ import annotations

import torch

from cusrl_amd.template.hook import Hook

__all__ = ["EmptyCudaCache"]


class EmptyCudaCache(Hook):
    """Returns the caching allocator's unused blocks to the device after each update — the reference does this
    unconditionally (``torch.cuda.empty_cache()`` in ``post_update``), the recurrent preset switches it on because BPTT
    minibatches of varying sequence counts fragment the cache on 16-80 GB devices.

    ``min_reserved_fraction`` (extension): skip the release while the allocator holds less than this fraction of the
    device's memory.  An MI355X has 288 GB; config 4 reserves ~8 GB, and giving that back every update only buys a round
    of hipFree / hipMalloc (and a device synchronisation) per iteration.  ``0.0`` is the reference's behaviour; captured
    hipGraphs keep their private pool either way."""

    def __init__(self, min_reserved_fraction: float = 0.5):
        if not 0.0 <= min_reserved_fraction <= 1.0:
            raise ValueError("'min_reserved_fraction' must be within [0, 1]")
        super().__init__()
        self.min_reserved_fraction = min_reserved_fraction
        self.releases = 0

    def post_update(self):
        if not torch.cuda.is_available():
            return
        device = self.agent.device if self.agent.device.type == "cuda" else None
        if self.min_reserved_fraction > 0.0:
            total = torch.cuda.get_device_properties(device).total_memory
            if torch.cuda.memory_reserved(device) < self.min_reserved_fraction * total:
                return
        torch.cuda.empty_cache()
        self.releases += 1
