"""Data-parallel collectives of the hot path (counterpart of cusrl/utils/distributed.py:35-188).

Process model: one rank per MI355X (``torchrun``), envs sharded per rank, manual gradient averaging.  On
PyTorch-ROCm ``backend="nccl"`` is RCCL over the xGMI mesh.  Every message here is latency-bound (<= 5 MB,
SURVEY.md §5), so the design goal is the fewest, copy-free collectives:

* ``reduce_gradients`` all-reduces ONE flat fp32 buffer.  With a :class:`FlatGradients` view installed
  (``ActorCritic`` does this) the parameters' ``.grad`` tensors alias that buffer, so there is no ``cat`` before
  and no copy-back after the collective (the reference does both, distributed.py:153-161).
* ``reduce_mean_var_`` all-gathers ``cat(mean, var)`` (8 bytes per rank for D = 1) and merges with the
  reference's equal-weight formula (distributed.py:175-183) — as a HIP kernel on device tensors.
"""

from __future__ import annotations

import os
from contextlib import contextmanager
from collections.abc import Iterable, Sequence
from typing import Any, TypeVar

import numpy as np
import torch

from cusrl_amd.utils.config import CONFIG, configure_distributed

__all__ = [
    "FlatGradients",
    "RcclComm",
    "native_comm",
    "average_dict",
    "barrier",
    "broadcast_parameters",
    "collective_route",
    "enabled",
    "gather_obj",
    "gather_stack",
    "is_main_process",
    "local_rank",
    "print_rank0",
    "rank",
    "reduce_gradients",
    "reduce_mean_",
    "reduce_mean_var_",
    "world_size",
]

_T = TypeVar("_T")


def enabled() -> bool:
    return CONFIG.distributed


def rank() -> int:
    return CONFIG.rank


def local_rank() -> int:
    return CONFIG.local_rank


def world_size() -> int:
    return CONFIG.world_size


def is_main_process() -> bool:
    return CONFIG.rank == 0


def print_rank0(*args, **kwargs):
    if CONFIG.rank == 0:
        print(*args, **kwargs)


_pg_stream: "torch.cuda.Stream | None" = None


@contextmanager
def _process_group_stream(device=None):
    """Issue a ``torch.distributed`` collective of an RCCL job on a dedicated stream that NEVER captures.

    ProcessGroupNCCL records a collective's completion event on the stream the call is issued on, and its watchdog thread
    keeps polling that event until its next wake-up after the event has fired (up to ~100 ms).  If the stream begins a
    hipGraph capture in that window — the agent's graph stream does, a few milliseconds after the eager warm-up of a step
    issued its gradient all-reduce there — the poll fails with "operation not permitted on an event last recorded in a
    capturing stream" and the watchdog aborts the process (seen as a sporadic SIGABRT of one-rank RCCL test jobs on the
    torch.distributed route, round 5).  So the process-group collectives of this package hop to their own stream: the
    caller's stream is joined before and after, i.e. the collective stays ordered exactly where it was issued.  Operands
    must be allocated by the caller (outside this context); what is allocated inside is consumed inside."""
    device = torch.device(CONFIG.device if device is None else device)
    if (device.type != "cuda" or torch.distributed.get_backend() != torch.distributed.Backend.NCCL
            or torch.cuda.is_current_stream_capturing()):
        yield
        return
    global _pg_stream
    if _pg_stream is None or _pg_stream.device != device:
        _pg_stream = torch.cuda.Stream(device=device)
    current = torch.cuda.current_stream(device)
    _pg_stream.wait_stream(current)
    with torch.cuda.stream(_pg_stream):
        yield
    current.wait_stream(_pg_stream)


def barrier():
    if configure_distributed():
        with _process_group_stream():
            torch.distributed.barrier()


class RcclComm:
    """A communicator created and used through the C ABI (``cusrl_comm_*`` in include/cusrl_hip.h, RCCL underneath).

    Rank 0 draws the id, the other ranks receive it over the existing torch.distributed group (control plane only);
    the collectives themselves are plain C calls that enqueue RCCL kernels on torch's current stream — capturable into
    a hipGraph, no Python-side work objects, no extra stream hops."""

    def __init__(self, world_size: int, rank: int, unique_id: bytes | None = None, device: torch.device | None = None):
        import ctypes

        from cusrl_amd import _native

        self._lib = lib = _native.lib()
        if not lib.cusrl_comm_available():
            raise _native.NativeError("RCCL is not available to libcusrl_hip.so: " + lib.cusrl_comm_last_error().decode())
        if unique_id is None:
            if world_size != 1:
                raise ValueError("a multi-rank communicator needs the id rank 0 drew (RcclComm.unique_id())")
            unique_id = self.unique_id()
        self.world_size, self.rank = world_size, rank
        self.device = torch.device(CONFIG.device if device is None else device)
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):  # ncclCommInitRank binds to the current device
            _native.check(lib.cusrl_comm_create(unique_id, world_size, rank, ctypes.byref(handle)), "cusrl_comm_create")
        self._handle = handle

    @staticmethod
    def unique_id() -> bytes:
        import ctypes

        from cusrl_amd import _native

        raw = ctypes.create_string_buffer(128)
        _native.check(_native.lib().cusrl_comm_unique_id(raw), "cusrl_comm_unique_id")
        return bytes(raw.raw)

    @classmethod
    def from_process_group(cls) -> "RcclComm":
        """Collective.  ``ncclCommInitRank`` blocks until every rank has joined, so nothing may make ONE rank leave before
        it: local preconditions (the library found RCCL, rank 0 could draw an id) are agreed on over the existing process
        group first, and only a unanimous yes proceeds to the collective creation."""
        from cusrl_amd import _native

        problem = ""
        try:
            if not _native.lib().cusrl_comm_available():
                problem = "RCCL is not available to libcusrl_hip.so: " + _native.lib().cusrl_comm_last_error().decode()
        except Exception as error:
            problem = f"{type(error).__name__}: {error}"
        payload: list = [None]
        if CONFIG.rank == 0 and not problem:
            try:
                payload[0] = cls.unique_id()
            except Exception as error:
                problem = f"{type(error).__name__}: {error}"
        with _process_group_stream():
            torch.distributed.broadcast_object_list(payload, src=0)  # None when rank 0 could not draw an id
        if payload[0] is None and not problem:
            problem = "rank 0 could not draw a communicator id"
        ready = torch.tensor([1.0 if problem else 0.0], device=CONFIG.device)
        with _process_group_stream():
            torch.distributed.all_reduce(ready, op=torch.distributed.ReduceOp.MAX)
        if ready.item() > 0:
            raise _native.NativeError(problem or "another rank cannot create its communicator")
        return cls(CONFIG.world_size, CONFIG.rank, payload[0])

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def _check(self, tensor: torch.Tensor, name: str):
        if not (tensor.is_cuda and tensor.is_contiguous() and tensor.device == self.device):
            raise ValueError(f"{name}: expected a contiguous tensor on {self.device}")

    def allreduce_mean_(self, tensor: torch.Tensor) -> torch.Tensor:
        from cusrl_amd import _native

        self._check(tensor, "allreduce_mean_")
        if tensor.dtype != torch.float32:
            raise TypeError("allreduce_mean_: float32 only (the flat gradient buffer)")
        _native.check(self._lib.cusrl_allreduce_mean(tensor.data_ptr(), tensor.numel(), self._handle, self._stream()),
                      "cusrl_allreduce_mean")
        return tensor

    def allgather(self, tensor: torch.Tensor) -> torch.Tensor:
        from cusrl_amd import _native

        self._check(tensor, "allgather")
        out = tensor.new_empty((self.world_size,) + tuple(tensor.shape))
        _native.check(self._lib.cusrl_allgather(tensor.data_ptr(), out.data_ptr(), tensor.numel() * tensor.element_size(),
                                                self._handle, self._stream()), "cusrl_allgather")
        return out

    def broadcast_(self, tensor: torch.Tensor, root: int = 0) -> torch.Tensor:
        from cusrl_amd import _native

        self._check(tensor, "broadcast_")
        _native.check(self._lib.cusrl_broadcast(tensor.data_ptr(), tensor.numel() * tensor.element_size(), root,
                                                self._handle, self._stream()), "cusrl_broadcast")
        return tensor

    def close(self):
        if getattr(self, "_handle", None):
            self._lib.cusrl_comm_destroy(self._handle)
            self._handle = None

    def abort(self):
        """Abandon the communicator without waiting for collectives that may never complete (``cusrl_comm_abort``)."""
        if getattr(self, "_handle", None):
            self._lib.cusrl_comm_abort(self._handle)
            self._handle = None


_native_comm: RcclComm | None = None
_native_comm_failed: str | None = None
_branch_comm: RcclComm | None = None  # second communicator: the critic window's all-reduce on the branch stream (split route)


def _agree(problem: str, device) -> bool:
    """Collective over the PROCESS GROUP (never over the communicator under test): True when no rank reported a problem.
    Every rank calls this the same number of times in the same order, whatever happened to it locally."""
    verdict = torch.tensor([1.0 if problem else 0.0], device=device)
    with _process_group_stream(device):
        torch.distributed.all_reduce(verdict, op=torch.distributed.ReduceOp.MAX)
    return verdict.item() == 0


def establish_native_comm(factory, device, rank: int, world: int, capture_probe=None):
    """Create the C-ABI communicator and prove it, in stages every rank walks through TOGETHER — each stage ends in a
    verdict all-reduce over the process group, and a rank that failed locally still issues every process-group collective of
    the stage it is in, so that no two ranks are ever inside different collectives (the failure mode of an earlier version:
    one rank raising in its probe skipped the reference all-reduce its peers were waiting in):

      1. create      ``factory()`` (itself collective-safe: local preconditions agreed on before ncclCommInitRank)
      2. eager probe a 64-float ``allreduce_mean_`` ENQUEUED ON A SIDE STREAM (a half-issued collective must not block the
                     stream the process-group collectives synchronise with); verdict "every rank enqueued"; only then the
                     side stream is joined and the result compared with the closed form; verdict "every rank agrees"
      3. captured    the same all-reduce captured into a hipGraph (local; verdict "every rank captured"), replayed on a
                     side stream without a device-wide synchronisation (verdict "every rank enqueued its replay"), then
                     joined and compared (verdict)

    Returns ``(comm, "")`` or ``(None, reason)`` — the same outcome on every rank.  A communicator that may have a
    half-issued collective in flight is aborted (``cusrl_comm_abort``), never destroyed (destroy waits for its kernels).
    ``CUSRL_COMM_FAULT = "<stage>:<rank>"`` injects a failure (tests of exactly this protocol)."""
    fault_stage, _, fault_rank = os.environ.get("CUSRL_COMM_FAULT", "").partition(":")

    def faulty(stage: str) -> bool:
        return fault_stage == stage and fault_rank != "" and int(fault_rank) == rank

    # ---- stage 1: creation
    comm, problem = None, ""
    try:
        if faulty("create"):
            raise RuntimeError("injected fault (CUSRL_COMM_FAULT)")
        comm = factory()
    except Exception as error:
        problem = f"{type(error).__name__}: {error}"
    if not _agree(problem, device):
        if comm is not None:
            comm.close()  # nothing was ever enqueued on it
        return None, problem or "another rank could not create its communicator"
    # ---- stage 2: one eager all-reduce, enqueued on a side stream
    base = torch.arange(64, dtype=torch.float32, device=device)
    probe, expect = base * (rank + 1), base * ((world + 1) / 2)  # mean over ranks of (rank + 1)
    side = torch.cuda.Stream(device=device) if device.type == "cuda" else None
    try:
        if faulty("probe"):
            raise RuntimeError("injected fault (CUSRL_COMM_FAULT)")
        if side is not None:
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):
                comm.allreduce_mean_(probe)
        else:
            comm.allreduce_mean_(probe)
    except Exception as error:
        problem = f"{type(error).__name__}: {error}"
    if not _agree(problem, device):  # some rank never enqueued its half: the others' kernels would wait forever
        comm.abort()
        return None, problem or "another rank could not enqueue the probe all-reduce"
    if side is not None:
        torch.cuda.current_stream(device).wait_stream(side)
    getattr(comm, "complete", lambda: None)()  # (host-side communicators of the CPU tests perform the enqueued work here)
    if not torch.allclose(probe, expect, rtol=1e-6, atol=0):
        problem = "cusrl_allreduce_mean returned a wrong mean"
    if not _agree(problem, device):
        comm.close()
        return None, problem or "another rank saw a wrong all-reduce result"
    # ---- stage 3: what the route exists for — the all-reduce INSIDE a hipGraph.  Same discipline as stage 2: a rank whose
    # capture failed must not leave its peers inside a replay that waits for its half, so (a) capture — local, nothing is
    # enqueued — and agree, (b) replay on a side stream WITHOUT a device-wide synchronisation and agree that every rank
    # enqueued, (c) only then join the side stream and compare.
    if capture_probe is None and device.type == "cuda":
        capture_probe = _capture_allreduce_probe  # (the CPU tests of this protocol hand in a host-side stand-in)
    if capture_probe is not None:
        captured = None
        try:
            if faulty("capture"):
                raise RuntimeError("injected fault (CUSRL_COMM_FAULT)")
            captured = capture_probe(comm, rank, world)
        except Exception as error:
            problem = f"cusrl_allreduce_mean inside a hipGraph: {type(error).__name__}: {error}"
        if not _agree(problem, device):
            comm.abort()  # (nothing was replayed anywhere, but a capture that broke half-way leaves the handle in no state to trust)
            return None, problem or "another rank could not capture the all-reduce"
        graph, probe, expect, side = captured
        try:
            if faulty("replay"):
                raise RuntimeError("injected fault (CUSRL_COMM_FAULT)")
            if side is not None:
                side.wait_stream(torch.cuda.current_stream(device))
                with torch.cuda.stream(side):
                    graph.replay()
            else:
                graph.replay()
        except Exception as error:
            problem = f"replay of a captured cusrl_allreduce_mean: {type(error).__name__}: {error}"
        if not _agree(problem, device):  # some rank never replayed its half: the others' kernels would wait forever
            comm.abort()
            return None, problem or "another rank could not replay the captured all-reduce"
        if side is not None:
            torch.cuda.current_stream(device).wait_stream(side)
        getattr(comm, "complete", lambda: None)()
        if not torch.allclose(probe, expect, rtol=1e-6, atol=0):
            problem = "a captured cusrl_allreduce_mean replayed a wrong result"
        if not _agree(problem, device):
            comm.close()
            return None, problem or "another rank saw a wrong result from the captured all-reduce"
    return comm, ""


def native_comm() -> RcclComm | None:
    """The process-wide C-ABI communicator — the default route of an RCCL job (``CONFIG.native_collectives``).  Created
    on first use, collectively: every rank must reach its first use together, which happens in ``broadcast_parameters``
    at agent construction.  Creation is verified before anything depends on it (:func:`establish_native_comm`); a problem
    on ANY rank is logged once and the whole job uses torch.distributed's collectives instead.
    ``None`` = collectives go through torch.distributed."""
    global _native_comm, _native_comm_failed
    if not (CONFIG.native_collectives and CONFIG.device.type == "cuda" and configure_distributed()):
        return None
    if torch.distributed.get_backend() != torch.distributed.Backend.NCCL:
        return None
    if _native_comm is None and _native_comm_failed is None:
        comm, problem = establish_native_comm(RcclComm.from_process_group, CONFIG.device, CONFIG.rank, CONFIG.world_size)
        if comm is None:
            _native_comm_failed = problem
            print(f"\033[1;33mcusrl_amd: C-ABI RCCL communicator unavailable on rank {CONFIG.rank} ({_native_comm_failed}); "
                  "falling back to torch.distributed collectives (eager all-reduce between two graphs per step)\033[0m",
                  flush=True)
        else:
            _native_comm = comm
            if CONFIG.split_gradient_allreduce:
                # two collectives in flight on two streams need two communicators (RCCL serialises the operations of one);
                # created right here, i.e. at the same point on every rank; without it the split route reduces both
                # windows through the one communicator, one after the other
                global _branch_comm
                second, problem = establish_native_comm(RcclComm.from_process_group, CONFIG.device, CONFIG.rank, CONFIG.world_size)
                if second is None:
                    print(f"\033[1;33mcusrl_amd: second C-ABI communicator unavailable on rank {CONFIG.rank} ({problem}); the split "
                          "gradient all-reduce uses one communicator for both windows\033[0m", flush=True)
                _branch_comm = second
    return _native_comm


def branch_comm() -> RcclComm | None:
    """The communicator of the critic window's all-reduce (``CONFIG.split_gradient_allreduce``), else None."""
    native_comm()
    return _branch_comm


def _capture_allreduce_probe(comm: RcclComm, rank: int, world: int):
    """Capture a 64-float ``cusrl_allreduce_mean`` into a hipGraph on a side stream (local work: nothing is enqueued on the
    communicator until somebody replays the graph).  Returns ``(graph, probe, expected mean, side stream)``."""
    device = comm.device
    base = torch.arange(64, dtype=torch.float32, device=device)
    probe = base * (rank + 1)
    expect = base * ((world + 1) / 2)  # mean over ranks of (rank + 1)
    stream = torch.cuda.Stream(device=device)
    stream.wait_stream(torch.cuda.current_stream(device))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream, capture_error_mode="thread_local"):
        comm.allreduce_mean_(probe)
    torch.cuda.current_stream(device).wait_stream(stream)
    return graph, probe, expect, stream


def collective_route() -> str:
    """Human-readable name of the route the hot path's collectives take in this process (bench / logs)."""
    if not configure_distributed():
        return "none (single process)"
    backend = torch.distributed.get_backend()
    if native_comm() is not None:
        return "c-abi rccl (cusrl_allreduce_mean / cusrl_allgather captured inside the step graph)"
    if backend == torch.distributed.Backend.NCCL:
        reason = f"; c-abi route unavailable: {_native_comm_failed}" if _native_comm_failed else ""
        return "torch.distributed rccl (eager all-reduce between two graphs per step)" + reason
    return f"torch.distributed {backend} (host-staged; eager all-reduce between two graphs per step)"


_host_pg = None


def host_group():
    """The process group HOST values travel through (the trainer's per-iteration log, gathered Python objects): the default group
    when that is gloo, else a gloo group next to the RCCL one — created on first use, which every rank reaches at the same point
    (``configure_distributed`` calls it right behind ``init_process_group``).  Why not RCCL: a device collective of host scalars
    needs an upload, a kernel and a read-back, and on this stack anything a host thread WAITS for on a side stream queues behind
    the work parked on the default stream (`scripts/probe_pg_stream.py`: 83 ms behind an 83 ms kernel, whichever stream or
    priority) — the pipelined trainer reads a log while the next rollout runs there."""
    global _host_pg
    if torch.distributed.get_backend() == torch.distributed.Backend.GLOO:
        return None
    if _host_pg is None:
        _host_pg = torch.distributed.new_group(backend="gloo")
    return _host_pg


def gather_obj(obj: _T) -> list[_T]:
    if not configure_distributed():
        return [obj]
    out: list[Any] = [None] * CONFIG.world_size
    torch.distributed.all_gather_object(out, obj, group=host_group())
    return out


_LOG_SLOTS = 58  # values of one fixed-size vector (64 fp64 = 512 B): the trainer's per-iteration log has ~25 keys


def _average_same_keys(info: dict[str, float]) -> dict[str, float] | None:
    """ONE fixed-size all-reduce when every rank reports the same (at most 58) keys with plain numbers — the steady
    state of the trainer's per-iteration log — instead of ``all_gather_object`` (pickling, two collectives, host
    round trips).  The vector carries a 2 x 16-bit digest of the key set, the digest's squares, the key count and a
    participation flag in front of the values; equal sums of a digest and of its square over the ranks mean every rank
    sent the same digest.  ``None`` = some rank differs or cannot take part; every rank sees the same sums, so all of
    them fall back to the general path together."""
    import zlib

    keys = sorted(info)
    values = [info[key] for key in keys]
    plain = len(keys) <= _LOG_SLOTS and all(isinstance(v, (int, float)) and not isinstance(v, bool) for v in values)
    digest = zlib.crc32("\0".join(keys).encode()) if plain else 0xFFFFFFFF
    low, high = float(digest & 0xFFFF), float(digest >> 16)
    head = [low, high, low * low, high * high, float(len(keys)), 1.0 if plain else 0.0]
    body = [float(v) for v in values] if plain else []
    packed = torch.tensor(head + body + [0.0] * (_LOG_SLOTS - len(body)), dtype=torch.float64)  # host values: a host collective
    torch.distributed.all_reduce(packed, op=torch.distributed.ReduceOp.SUM, group=host_group())
    summed = packed.tolist()
    world = CONFIG.world_size
    s_low, s_high, q_low, q_high, count, all_plain = summed[:6]
    same = (all_plain == world and s_low == world * low and s_high == world * high and q_low == world * low * low
            and q_high == world * high * high and count == world * len(keys))
    if not same:
        return None
    return {key: value / world for key, value in zip(keys, summed[6:])}


def average_dict(info: dict[str, float]) -> dict[str, float]:
    """Rank-average every key that at least one rank reported (trainer.py:387)."""
    if not configure_distributed() or world_size() == 1:
        return info  # (a process group of one: the mean over the ranks is the rank's own value — no collective, no host copy)
    if (averaged := _average_same_keys(info)) is not None:
        return averaged
    gathered = gather_obj(info)
    keys = {key for item in gathered for key in item}
    result = {}
    for key in keys:
        values = [item[key] for item in gathered if item.get(key) is not None]
        if values:
            result[key] = float(np.mean(values))
    return result


def broadcast_parameters(parameters: Iterable[torch.nn.Parameter]):
    """Rank 0's parameters to every rank, as ONE flat broadcast instead of one per tensor."""
    if not configure_distributed():
        return
    params = [p for p in parameters]
    if not params:
        return
    flat = torch.cat([p.data.reshape(-1) for p in params])
    if (comm := native_comm()) is not None and flat.is_cuda:
        comm.broadcast_(flat, 0)
    else:
        with _process_group_stream(flat.device):
            torch.distributed.broadcast(flat, src=0)
    offset = 0
    for p in params:
        n = p.numel()
        p.data.copy_(flat[offset : offset + n].view_as(p.data))
        offset += n


def gather_stack(tensor: torch.Tensor) -> torch.Tensor:
    """``[W, *tensor.shape]`` with every rank's tensor."""
    if not configure_distributed():
        return tensor.unsqueeze(0)
    if tensor.is_cuda and (comm := native_comm()) is not None:
        return comm.allgather(tensor.contiguous())
    if torch.distributed.get_backend() == torch.distributed.Backend.GLOO:
        parts = [torch.empty_like(tensor) for _ in range(CONFIG.world_size)]
        torch.distributed.all_gather(parts, tensor)
        return torch.stack(parts, dim=0)
    out = tensor.new_empty(CONFIG.world_size, *tensor.shape)
    with _process_group_stream(tensor.device):
        torch.distributed.all_gather_into_tensor(out, tensor)
    return out


def reduce_mean_(tensor: torch.Tensor) -> torch.Tensor:
    """In-place cross-rank average (RCCL ``AVG``; Gloo has no AVG, so SUM then divide)."""
    if not configure_distributed():
        return tensor
    if tensor.is_cuda and tensor.dtype == torch.float32 and tensor.is_contiguous() and (comm := native_comm()) is not None:
        return comm.allreduce_mean_(tensor)
    if torch.distributed.get_backend() == torch.distributed.Backend.GLOO:
        torch.distributed.all_reduce(tensor, op=torch.distributed.ReduceOp.SUM)
        return tensor.div_(CONFIG.world_size)
    with _process_group_stream(tensor.device):
        torch.distributed.all_reduce(tensor, op=torch.distributed.ReduceOp.AVG)
    return tensor


def reduce_mean_var_(mean: torch.Tensor, var: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """Equal-weight merge of per-rank statistics into ``mean`` / ``var`` in place."""
    if not configure_distributed():
        return mean, var
    gathered = gather_stack(torch.cat((mean, var), dim=0))  # [W, 2D]
    if mean.is_cuda:
        from cusrl_amd import ops

        ops.merge_mean_var(gathered, mean, var)
    else:  # host-side statistics (CPU process groups in tests): same formula, same order
        all_means, all_vars = gathered.chunk(2, -1)
        torch.mean(all_means, dim=0, out=mean)
        torch.mean(all_vars + (all_means - mean).square(), dim=0, out=var)
    return mean, var


class FlatGradients:
    """One contiguous fp32 buffer that every parameter's ``.grad`` aliases.

    ``zero()`` replaces ``optimizer.zero_grad()`` with a single memset and keeps the aliases alive, so the
    per-step gradient all-reduce is one collective on one registered buffer with no packing traffic.
    """

    def __init__(self, optimizer: torch.optim.Optimizer):
        self.params = [p for group in optimizer.param_groups for p in group["params"] if p.requires_grad]
        device = self.params[0].device
        # every parameter's window starts on a 16-byte boundary (4 floats): the kernels that read weights / write
        # gradients with 16-byte lanes (narrow-head backward, assembly, Adam) need it, and a [1]-element bias in front
        # would otherwise shift everything behind it.  The padding slots stay zero (zero gradient -> Adam leaves them 0).
        self.offsets: list[int] = []
        total = 0
        for p in self.params:
            self.offsets.append(total)
            total += -(-p.numel() // 4) * 4
        self.buffer = torch.zeros(total, dtype=torch.float32, device=device)
        self.views = []
        self.absent: list[int] = []
        # the per-network split of the backward (ActorCritic._backward): the windows it reduces, and whether it already did
        self.split_windows: list[torch.Tensor] | None = None
        self.reduced = False
        # a backward that assembled the critic's window on the critic's stream and the others' on the main stream WITHOUT joining
        # them (ActorCritic._backward, round 6): events, squared-norm rows and element ranges for the two-window optimizer step
        self.split_tail: dict | None = None
        self._sumsq: torch.Tensor | None = None
        self._sumsq_version = -1
        for p, offset in zip(self.params, self.offsets):
            if p.dtype != torch.float32:
                raise TypeError("FlatGradients expects fp32 master parameters")
            self.views.append(self.buffer[offset : offset + p.numel()].view_as(p))
        self.attach()

    def packed(self) -> torch.Tensor:
        """The gradients as the reference's ``torch.cat`` would give them (windows without their alignment padding)."""
        return torch.cat([view.reshape(-1) for view in self.views])

    def attach(self):
        for p, view in zip(self.params, self.views):
            p.grad = view

    def intact(self) -> bool:
        return all(p.grad is view for p, view in zip(self.params, self.views))

    def zero(self):
        if not self.intact():
            self.attach()
        self.buffer.zero_()

    def assemble(self, grads, split_slabs: dict[int, torch.Tensor] | None = None, subset: Sequence[int] | None = None,
                 want_sumsq: bool | None = None):
        """Write every parameter's gradient into its slot with ONE launch (``cusrl_assemble_gradients``).
        ``grads[i]`` is parameter i's gradient or None; ``split_slabs`` maps a parameter's storage address to the
        unsummed ``[S, ...]`` partial gradients its split-batch GEMM left behind (cusrl_amd/nn/module.py).
        ``subset`` (indices into ``params``, ``grads`` aligned with it): only those parameters' windows are written — the
        per-network split of the backward (``ActorCritic._backward``) assembles critic and actor windows separately; slabs
        of parameters outside the subset (the shared loss node hands the std vector's over in both passes) are dropped.
        ``want_sumsq`` (default: a single process assembling everything): also return the blocks' partial sums of squares."""
        from cusrl_amd import ops

        pieces = []
        if subset is None:
            params, offsets = self.params, self.offsets
        else:
            params, offsets = [self.params[i] for i in subset], [self.offsets[i] for i in subset]
            mine = {p.data_ptr() for p in params}
            others = {p.data_ptr() for p in self.params} - mine
            # slabs of the OTHER windows' parameters (a loss node both passes share hands the std vector's over in each) are
            # that pass's business; a key that belongs to no optimizer parameter at all stays and raises below
            for key in [key for key in (split_slabs or {}) if key in others]:
                del split_slabs[key]
        # parameters autograd returned no gradient for (unused this step): torch's optimizers skip them; the flat Adam
        # step reads this list and leaves their windows untouched (utils/flat_optimizer.py)
        absent = [(i if subset is None else subset[i]) for i, (p, grad) in enumerate(zip(params, grads))
                  if grad is None and not (split_slabs and p.data_ptr() in split_slabs)]
        # (a split backward assembles window by window: the caller empties `absent` first, every window adds its own)
        self.absent = absent if subset is None else sorted(set(self.absent) | set(absent))
        for p, grad, offset in zip(params, grads, offsets):
            n = p.numel()
            slabs = split_slabs.pop(p.data_ptr(), None) if split_slabs else None
            pending = isinstance(slabs, ops.DeferredColumns)  # partial rows of a column-sum kernel
            if slabs is not None and grad is not None:  # the parameter was used more than once in the graph
                grad, slabs = grad + (slabs.materialize() if pending else ops.sum_slabs(slabs)).view_as(grad), None
            if slabs is not None:
                pieces.append((slabs, offset, n, slabs.splits if pending else slabs.shape[0]))
            elif grad is not None:
                pieces.append((grad, offset, n, 1))
            else:
                pieces.append((None, offset, n, 0))
        if split_slabs:
            raise RuntimeError("split weight gradients were produced for tensors that are not optimizer parameters")
        # single process: nothing changes the gradients between here and the clipping, so the assembly also leaves the
        # partial sums of squares the clipping coefficient needs (with several ranks the all-reduce comes in between)
        if want_sumsq is None:
            want_sumsq = not configure_distributed() and subset is None
        sumsq = ops.assemble_gradients(pieces, self.buffer, want_sumsq=want_sumsq)
        self._sumsq = sumsq if subset is None else None
        self._sumsq_version = self.buffer._version
        return sumsq

    def element_range(self, indices: Sequence[int]) -> tuple[int, int]:
        """``(first, end)`` element offsets of the run of consecutive parameters ``indices`` (padding included)."""
        first, last = indices[0], indices[-1]
        if list(indices) != list(range(first, last + 1)):
            raise ValueError("a gradient window is a run of consecutive parameters")
        return self.offsets[first], (self.offsets[last + 1] if last + 1 < len(self.offsets) else self.buffer.numel())

    def window(self, indices: Sequence[int]) -> torch.Tensor:
        """The contiguous stretch of the buffer that holds the (consecutive) parameters ``indices``, padding included."""
        first, last = indices[0], indices[-1]
        if list(indices) != list(range(first, last + 1)):
            raise ValueError("a gradient window is a run of consecutive parameters")
        end = self.offsets[last + 1] if last + 1 < len(self.offsets) else self.buffer.numel()
        return self.buffer[self.offsets[first] : end]

    def take_sumsq(self) -> torch.Tensor | None:
        """The squared-norm partials the last :meth:`assemble` produced, if the gradients are still what it wrote (any
        in-place edit through a ``.grad`` view moves the buffer's version counter)."""
        sumsq, self._sumsq = self._sumsq, None
        return sumsq if sumsq is not None and self.buffer._version == self._sumsq_version else None


def reduce_gradients(optimizer: torch.optim.Optimizer, flat: FlatGradients | None = None):
    """Average gradients across ranks (actor_critic.py:314)."""
    if not configure_distributed():
        return
    if flat is not None and flat.intact():
        flat._sumsq = None  # the averaged gradients have another norm
        tail = flat.split_tail
        if tail is not None and tail.get("reduce"):
            # An unjoined step (ActorCritic._backward): the critic's window was assembled on the critic's stream, the others' on
            # this one.  ONE collective over the whole buffer, here, behind the critic's assembly; the critic's step launch waits
            # for it through the event its stream would have waited for anyway (FlatAdam.step: `main_assembled`) — the step still
            # has two cross-stream edges and no join.
            main = torch.cuda.current_stream()
            main.wait_event(tail["branch_assembled"])
            reduce_mean_(flat.buffer)
            averaged = torch.cuda.Event()
            averaged.record(main)
            tail.update(main_assembled=averaged, main_joined=True, reduced=True, reduce=False)
            return
        if flat.reduced:  # the split route already averaged both windows inside the backward
            flat.reduced = False
            return
        windows = flat.split_windows
        if CONFIG.split_gradient_allreduce and windows:
            # the split route where the collectives cannot be issued inside the backward (a captured phase on a route that
            # cannot be captured: torch.distributed's collectives): the same windows, one after the other, here
            for window in windows:
                reduce_mean_(window)
            return
        reduce_mean_(flat.buffer)
        return
    params = [p for group in optimizer.param_groups for p in group["params"] if p.grad is not None]
    if not params:
        return
    grads = torch.cat([p.grad.reshape(-1) for p in params])
    reduce_mean_(grads)
    offset = 0
    for p in params:
        n = p.numel()
        p.grad.copy_(grads[offset : offset + n].view_as(p.grad))
        offset += n
