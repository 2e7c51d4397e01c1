"""One RCCL rank (torchrun, backend nccl == RCCL) running the ppo preset: exercises parameter broadcast, per-step flat
gradient all-reduce (eager between graph replays, or captured inside the step's hipGraph when the collectives go through
the C ABI), advantage-statistics all-gather + HIP merge, metric averaging.  ``argv``: out-dir, compile (0/1), native (0/1)."""

import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

import cusrl_amd as cusrl  # noqa: E402
from cusrl_amd import _native  # noqa: E402
from cusrl_amd.utils import distributed  # noqa: E402


def main(out_dir: str, compile_: str, native: str):
    assert distributed.enabled()
    cusrl.config.native_collectives = native == "1"
    cusrl.utils.configure_distributed()
    shared = cusrl.config.share_gpu  # test-only: every rank on cuda:0, gloo process group (CUSRL_SHARE_GPU=1)
    assert torch.distributed.get_backend() == ("gloo" if shared else "nccl")
    assert not shared or torch.cuda.current_device() == 0
    rank, world = distributed.rank(), distributed.world_size()
    cusrl.set_global_seed(5)
    env = cusrl.testing.SyntheticEnvironment(256, 20, 6)
    factory = cusrl.preset.PpoAgentFactory(num_steps_per_update=8, sampler_epochs=3, sampler_mini_batches=2,
                                           compile=compile_ == "1")
    trainer = cusrl.Trainer(env, factory, num_iterations=6, verbose=False)  # (the whole-update graph is captured in the fourth)
    first_perm = torch.randperm(16, device="cuda").tolist()  # per-rank generator streams differ (seed + rank)
    trainer.run_training_loop()
    params = torch.cat([p.detach().reshape(-1) for p in trainer.agent.parameters()])
    # the advantage statistics every rank normalised with: local (mean, var) -> all_gather -> HIP merge
    local_mean = torch.tensor([1.0 + rank, -2.0 * rank], device="cuda")
    local_var = torch.tensor([0.5 + rank, 2.0], device="cuda")
    mean, var = local_mean.clone(), local_var.clone()
    distributed.reduce_mean_var_(mean, var)
    flat = torch.full((1000,), float(rank + 1), device="cuda")
    distributed.reduce_mean_(flat)
    info = {k: v for k, v in trainer.last_info.items() if k.startswith("Agent/")}
    # the trainer's log average as a multi-rank job issues it (`average_dict` answers a group of one without a collective): host
    # values through the job's gloo group (`distributed.host_group`) — with a long kernel parked on the device's default stream, as
    # the next rollout is when the pipelined trainer reads a log (round 6): it must not wait for the device
    torch.cuda._sleep(200_000_000)  # ~80 ms on the current stream
    probe = {"Agent/a": 1.5 + rank, "Metric/b": -2.0, "Perf/c": 3}
    began = __import__("time").perf_counter()
    averaged = distributed._average_same_keys(dict(probe))
    log_average_s = __import__("time").perf_counter() - began
    expected = {"Agent/a": 1.5 + (world - 1) / 2, "Metric/b": -2.0, "Perf/c": 3.0}
    assert averaged is not None and all(abs(averaged[k] - expected[k]) < 1e-12 for k in expected), averaged
    torch.cuda.synchronize()
    graphs = [step.single_graph for step in getattr(trainer.agent, "_graphed_steps", {}).values()]
    Path(out_dir, f"rank{rank}.json").write_text(json.dumps({
        "info": info, "world": world, "rank": rank, "first_perm": first_perm,
        "local_mean": local_mean.tolist(), "local_var": local_var.tolist(), "mean": mean.tolist(), "var": var.tolist(),
        "flat_mean": flat[0].item(), "param_sum": params.double().sum().item(), "param_head": params[:64].tolist(),
        "param_bytes": params.cpu().numpy().tobytes().hex()[:4096],
        "native": distributed.native_comm() is not None, "single_graph": graphs, "route": distributed.collective_route(),
        "captured_env_steps": trainer._graphed_rollout.captured if trainer._graphed_rollout is not None else 0,
        "advantage_head": trainer.agent.buffer["advantage"].flatten()[:8].tolist(),
        "split_backward": bool(trainer.agent._split_plan),
        "allreduce_calls": _native.launch_counts.get("cusrl_allreduce_mean", 0),
        "allgather_calls": _native.launch_counts.get("cusrl_allgather", 0),
        "log_average_s": log_average_s,
        "update_graph_replays": getattr(getattr(trainer.agent, "_graphed_epochs", None), "replays", 0),
        "two_window_steps": getattr(trainer.agent.flat_optimizer, "two_window_steps", 0),
        "normed_steps": _native.launch_counts.get("cusrl_adam_step_normed", 0),
        "sumsq_launches": _native.launch_counts.get("cusrl_grad_sumsq", 0),
    }))
    distributed.barrier()
    torch.cuda.synchronize()
    # The results are on disk and every rank is past the last collective: say so, then leave WITHOUT tearing the process group
    # down.  torch.distributed.destroy_process_group() (and, worse, interpreter exit) races ProcessGroupNCCL's watchdog thread —
    # it polls its works' events while the communicator and the events are being destroyed; once in a few dozen runs a HIP
    # call of its loop fails and the process aborts after all the work is done.  os._exit ends the threads with the process.
    print(f"WORKER_RESULTS_WRITTEN rank {rank}", flush=True)
    sys.stdout.flush(), sys.stderr.flush()
    import os

    os._exit(0)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
