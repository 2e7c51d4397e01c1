"""Process-wide device / rank configuration (counterpart of cusrl/utils/config.py:13-200).

One process drives one MI355X.  Rank, local rank and world size come from the ``torchrun`` environment
(``RANK`` / ``LOCAL_RANK`` / ``WORLD_SIZE``); on PyTorch-ROCm the ``nccl`` backend IS RCCL, so collectives
run over xGMI; CPU-only processes (tests) use ``gloo``.
"""

from __future__ import annotations

import atexit
import os

import torch

__all__ = ["CONFIG", "configure_distributed", "device", "is_autocast_available"]


class _Config:
    def __init__(self):
        self.cuda = torch.cuda.is_available()
        self.seed: int | None = None
        env = os.environ
        self.distributed = "LOCAL_RANK" in env
        self.rank = int(env.get("RANK", 0)) if self.distributed else 0
        self.local_rank = int(env.get("LOCAL_RANK", 0)) if self.distributed else 0
        self.world_size = int(env.get("WORLD_SIZE", 1)) if self.distributed else 1
        self.local_world_size = int(env.get("LOCAL_WORLD_SIZE", self.world_size)) if self.distributed else 1
        # Test-only: every rank of the job drives cuda:0 and the process group is gloo (RCCL refuses two ranks on one
        # device; gloo stages device tensors through the host).  Lets a 1-GPU box run the whole multi-rank agent path —
        # parameter broadcast, per-step gradient averaging, merged advantage statistics, rank-averaged logs — end to end.
        self.share_gpu = self.distributed and env.get("CUSRL_SHARE_GPU", "0") == "1"
        self._device = torch.device((f"cuda:{0 if self.share_gpu else self.local_rank}") if self.cuda else "cpu")
        # Collectives of the hot path through the C ABI (cusrl_allreduce_mean / cusrl_allgather / cusrl_broadcast on a
        # communicator owned by libcusrl_hip.so) instead of torch.distributed: they are enqueued on the step's stream,
        # so with compile=True the gradient all-reduce is captured INSIDE the minibatch step's hipGraph.  This is the
        # DEFAULT route of an RCCL job (utils/distributed.py native_comm: the communicator is created over the existing
        # process group and checked against it once; a failure is logged and the job falls back to torch.distributed's
        # collectives).  CUSRL_NATIVE_COLLECTIVES=0 (or CONFIG.native_collectives = False before the agent is built)
        # forces the torch.distributed route: eager all-reduce between two graphs per minibatch step.
        self.native_collectives = env.get("CUSRL_NATIVE_COLLECTIVES", "1") != "0"
        # Per-network split of the gradient all-reduce (cusrl/utils/distributed.py:145-172 is ONE all-reduce behind the whole
        # backward): the critic's parameters are differentiated first, their window of the flat buffer is assembled and
        # averaged on the branch stream through a second communicator WHILE the actor's backward runs; the actor's window
        # follows on the main stream.  Same kernels, same operands: bit-identical parameters (tests/test_distributed_*).
        # Off by default — whether two half-size collectives overlapped with backward beat one full-size collective behind
        # it on 8 xGMI-connected ranks has never been measured (no multi-GPU box in any round); bench.py prints both routes'
        # durations so that the first such session is one A/B.  CUSRL_SPLIT_ALLREDUCE=1 or CONFIG.split_gradient_allreduce.
        self.split_gradient_allreduce = env.get("CUSRL_SPLIT_ALLREDUCE", "0") == "1"
        # Building an agent on a GPU loads the measured rocBLAS / hipBLASLt kernel selection through PyTorch TunableOp
        # (utils/tuning.py) — a PROCESS-WIDE setting: other torch code in the process gets the same kernel choice for
        # the GEMM shapes listed in the file.  CONFIG.tuned_gemms = False before the first agent is built (or
        # CUSRL_TUNED_GEMMS=0) leaves torch untouched.
        self.tuned_gemms = env.get("CUSRL_TUNED_GEMMS", "1") != "0"

    @property
    def device(self) -> torch.device:
        return self._device

    @device.setter
    def device(self, value):
        self._device = torch.device(value)

    def set_device(self, value):
        self.device = value


CONFIG = _Config()


def device(device: str | torch.device | None = None) -> torch.device:
    """The given device, or the process default when ``None``."""
    return CONFIG.device if device is None else torch.device(device)


def is_autocast_available() -> bool:
    return CONFIG.cuda and torch.amp.autocast_mode.is_autocast_available(CONFIG.device.type)


def configure_distributed(backend: str | None = None, **kwargs) -> bool:
    """Lazily create the default process group; returns whether this is a multi-process job."""
    if not CONFIG.distributed:
        return False
    if not torch.distributed.is_initialized():
        if backend is None:
            backend = "nccl" if CONFIG.device.type == "cuda" and not CONFIG.share_gpu else "gloo"  # nccl == RCCL on ROCm
        if CONFIG.device.type == "cuda":
            torch.cuda.set_device(CONFIG.device)
            if backend == "nccl":
                kwargs.setdefault("device_id", CONFIG.device)
        torch.distributed.init_process_group(backend=backend, world_size=CONFIG.world_size, rank=CONFIG.rank, **kwargs)
        from cusrl_amd.utils.distributed import host_group

        host_group()  # the gloo group of an RCCL job's host values (the trainer's log): created where every rank passes together
    return True


@atexit.register
def _shutdown():
    if CONFIG.distributed and torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
