"""Worker for tests/test_distributed_cpu.py: one rank of a world_size-2 gloo job (launched by torchrun)."""

import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

import cusrl_amd as cusrl  # noqa: E402  (reads RANK / LOCAL_RANK / WORLD_SIZE at import)
from cusrl_amd.utils import distributed  # noqa: E402


def main(out_dir: str):
    cusrl.config.set_device("cpu")
    assert distributed.enabled() and distributed.world_size() == 2
    rank = distributed.rank()
    result = {"rank": rank}

    # per-rank seeding: seed + rank (cusrl/utils/misc.py:163)
    cusrl.set_global_seed(7)
    result["first_randperm"] = torch.randperm(16).tolist()

    # a6: advantage statistics merge across ranks (equal-weight formula)
    mean = torch.tensor([1.0 + rank, -2.0 * rank])
    var = torch.tensor([0.5 + rank, 2.0])
    result["local_mean"], result["local_var"] = mean.tolist(), var.tolist()
    distributed.reduce_mean_var_(mean, var)
    result["merged_mean"], result["merged_var"] = mean.tolist(), var.tolist()

    # a14: flat gradient averaging, with and without the aliasing flat buffer
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.Linear(4, 2))
    optimizer = torch.optim.SGD(model.parameters(), lr=0.1)
    flat = distributed.FlatGradients(optimizer)
    flat.zero()
    loss = model(torch.full((5, 3), float(rank + 1))).square().sum()
    loss.backward()
    assert flat.intact()
    local = flat.packed()
    distributed.reduce_gradients(optimizer, flat)
    result["flat_local"], result["flat_reduced"] = local.tolist(), flat.packed().tolist()
    for p in model.parameters():
        p.grad = None
    model(torch.full((5, 3), float(rank + 1))).square().sum().backward()
    distributed.reduce_gradients(optimizer)  # reference-style cat / all-reduce / copy-back path
    result["cat_reduced"] = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).tolist()

    # parameter broadcast from rank 0
    torch.manual_seed(100 + rank)
    other = torch.nn.Linear(3, 2)
    before = torch.cat([p.detach().reshape(-1) for p in other.parameters()]).tolist()
    distributed.broadcast_parameters(other.parameters())
    result["params_before"] = before
    result["params_after"] = torch.cat([p.detach().reshape(-1) for p in other.parameters()]).tolist()

    result["averaged"] = distributed.average_dict({"shared": float(rank), f"only{rank}": 1.0})  # different keys: gather path
    result["averaged_same_keys"] = distributed.average_dict({"b": float(rank), "a": 10.0 * rank, "c": 3})  # all-reduce path
    result["fast_path_taken"] = distributed._average_same_keys({"x": float(rank)}) is not None
    result["averaged_with_none"] = distributed.average_dict({"a": 1.0 + rank, "b": None if rank else 2.0})
    result["gathered"] = distributed.gather_obj(rank * 10)
    result["stack"] = distributed.gather_stack(torch.tensor([float(rank)])).tolist()
    # KL-driven learning-rate control: every rank acts on the MEAN KL over ranks (lr_schedule.py:60-62, 287-296), so the
    # ranks, fed different KLs, must take the same decisions
    sys.path.insert(0, str(ROOT / "tests" / "golden"))
    from make_golden import ScheduleProbe  # noqa: PLC0415

    kls = [0.004 + 0.002 * rank, 0.05 - 0.04 * rank, 0.011, 0.0001 + 0.03 * rank]
    rows = ScheduleProbe().run(cusrl.hook.ThresholdLRSchedule(desired_kl_divergence=0.01), kls, False)
    result["threshold_lrs"] = rows[:, 0].tolist()
    rows = ScheduleProbe().run_mini_batch_wise(cusrl.hook.MiniBatchWiseLRSchedule(desired_kl_divergence=0.01),
                                               cusrl.hook.OnPolicyPreparation(), kls, 2)
    result["mini_batch_wise_lrs"] = rows[:, 0].tolist()
    # start-up protocol of the C-ABI communicator (utils/distributed.py establish_native_comm) with a host-side stand-in
    # for the communicator: every stage must end with the SAME outcome on both ranks and leave the process group's
    # collective sequence aligned, whichever rank fails at whichever stage
    class HostComm:
        """Enqueue-then-complete like a stream-ordered RCCL call: `allreduce_mean_` only records the request."""

        device, world_size = torch.device("cpu"), 2
        closed = aborted = False

        def __init__(self):
            self.pending = []

        def allreduce_mean_(self, tensor):
            self.pending.append(tensor)

        def complete(self):
            for tensor in self.pending:
                torch.distributed.all_reduce(tensor)
                tensor.div_(2)
            self.pending.clear()

        def close(self):
            self.closed = True

        def abort(self):
            self.aborted = True

    class HostGraph:
        """Stand-in for the captured all-reduce: capturing enqueues nothing, a replay enqueues the collective."""

        def __init__(self, comm, probe):
            self.comm, self.probe = comm, probe

        def replay(self):
            self.comm.allreduce_mean_(self.probe)

    def host_capture(comm, rank_, world):
        base = torch.arange(64, dtype=torch.float32)
        probe = base * (rank_ + 1)
        return HostGraph(comm, probe), probe, base * ((world + 1) / 2), None

    outcomes = {}
    for scenario in ("", "create:0", "create:1", "probe:0", "probe:1", "capture:0", "capture:1", "replay:0", "replay:1"):
        os.environ["CUSRL_COMM_FAULT"] = scenario
        created = []

        def factory():
            created.append(HostComm())
            return created[-1]

        comm, reason = distributed.establish_native_comm(factory, torch.device("cpu"), rank, 2, capture_probe=host_capture)
        check = torch.tensor([float(rank + 1)])
        torch.distributed.all_reduce(check)  # the process group is still in step: 1 + 2
        outcomes[scenario or "none"] = {"ok": comm is not None, "reason": reason, "in_step": check.item() == 3.0,
                                        "closed": bool(created and created[-1].closed), "aborted": bool(created and created[-1].aborted)}
    os.environ.pop("CUSRL_COMM_FAULT", None)
    result["comm_protocol"] = outcomes
    distributed.barrier()
    Path(out_dir, f"rank{rank}.json").write_text(json.dumps(result))


if __name__ == "__main__":
    main(sys.argv[1])
