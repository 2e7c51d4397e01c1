#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the rollout + PPO-update loop (BASELINE.json `metric`, config 2).

One "step" = one training iteration = 24 vectorised env steps of 4096 synthetic envs (obs 48, act 12) pushed into
the HBM rollout buffer + one full `ppo`-preset update (next_value, GAE, advantage normalisation, 5 epochs x 4
minibatches of gather + fused PPO objective + backward + Adam, statistics pass).  Data are synthetic and resident
on the GPU; weights are random-init.  Multi-GPU (launched by torchrun, one rank per GPU): envs are sharded, 4096
per rank (weak scaling), gradients and advantage statistics are all-reduced over RCCL/xGMI.

Prints ONE JSON line on rank 0.  Extra objects:
  roofline     — dominant hot-path kernel (minibatch gather): algorithmic bytes / HIP-event time vs 8 TB/s HBM
  kernels      — same accounting for every HIP kernel of the path (separate short instrumented pass)
  cpu_baseline — oracle/torch_ppo.py (reference-equivalent torch CPU path) timed on this box's host cores
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
NUM_ENVS, OBS_DIM, ACT_DIM, HORIZON = 4096, 48, 12, 24


def parse_args():
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--steps", type=int, default=40)
    parser.add_argument("--warmup", type=int, default=8)
    parser.add_argument("--envs-per-gpu", type=int, default=NUM_ENVS)
    parser.add_argument("--no-cpu-baseline", action="store_true")
    parser.add_argument("--no-kernel-pass", action="store_true")
    parser.add_argument("--cpu-seconds", type=float, default=15.0)
    parser.add_argument("--eager", action="store_true", help="disable hipGraph replay (compile=False)")
    return parser.parse_args()


class EventRecorder:
    """HIP-event pairs around calls of named ``cusrl_amd.ops`` functions, recorded on torch's current stream (the
    stream every kernel of the path is launched on)."""

    def __init__(self):
        self.pending: dict[str, list] = {}
        self.bytes: dict[str, float] = {}
        self.originals: dict[str, object] = {}

    def wrap(self, ops, name, bytes_fn):
        original = getattr(ops, name)
        self.originals[name] = original
        pending = self.pending.setdefault(name, [])

        def timed(*args, **kwargs):
            start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
            result = original(*args, **kwargs)
            end.record()
            pending.append((start, end, bytes_fn(*args, **kwargs)))
            return result

        setattr(ops, name, timed)

    def unwrap(self, ops):
        for name, original in self.originals.items():
            setattr(ops, name, original)
        self.originals.clear()

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, records in self.pending.items():
            if not records:
                continue
            total_ms = sum(s.elapsed_time(e) for s, e, _ in records)
            total_bytes = sum(b for _, _, b in records)
            avg_us = total_ms * 1e3 / len(records)
            gbs = total_bytes / (total_ms * 1e-3) / 1e9 if total_ms > 0 else 0.0
            out[name] = {
                "launches": len(records),
                "avg_us": round(avg_us, 3),
                "bytes_per_launch": int(total_bytes / len(records)),
                "achieved_GBps": round(gbs, 1),
                "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4),
            }
        return out


def _row_bytes(t, lead):
    n = t.element_size()
    for s in t.shape[lead:]:
        n *= s
    return n


# algorithmic bytes of one call (SURVEY.md §8d accounting: every byte the op must read + write once)
def gather_bytes(storages, indices, capacity, parallelism, temporal=False):
    rows = indices.numel() * (capacity if temporal else 1)
    return rows * sum(2 * _row_bytes(s, 2) for s in storages) + indices.numel() * 8


def push_bytes(pairs, cursor, parallelism):
    return sum(2 * step.numel() * step.element_size() for step, _ in pairs)


def gae_bytes(reward, value, next_value, done, *a, **k):
    return reward.numel() * (3 * 4 + 2 * 4) + done.numel()


def next_value_bytes(value, terminated, truncated, *a, **k):
    return value.numel() * 8 + terminated.numel() * 2


def normalize_bytes(x, *a, **k):
    return x.numel() * 8


def loss_bytes(advantage, old_logp, action, mean, std, ret, curr_value, old_value, **k):
    B, A, D = advantage.numel(), mean.shape[-1], ret.shape[-1]
    return B * (4 + 4 + 3 * 4 * A + 2 * 4 * D + 2 * 4 * A + 4 * D + 4 * 4)  # + logp/entropy/ratio side outputs


def run_gpu(args, rank, world):
    import cusrl_amd as cusrl
    from cusrl_amd import ops
    from cusrl_amd.utils import distributed

    device = torch.device(f"cuda:{int(os.environ.get('LOCAL_RANK', 0))}")
    torch.cuda.set_device(device)
    cusrl.config.set_device(device)
    if world > 1:
        cusrl.utils.configure_distributed()
    cusrl.set_global_seed(42)
    env = cusrl.testing.SyntheticEnvironment(args.envs_per_gpu, OBS_DIM, ACT_DIM, device=device)
    # compile=True = hipGraph replay of the act step and the minibatch steps (cusrl_amd/template/graphs.py)
    factory = cusrl.preset.PpoAgentFactory(compile=not args.eager, optimizer_kwargs={"fused": True, "capturable": True})
    trainer = cusrl.Trainer(env, factory, num_iterations=10**9, verbose=False)
    agent = trainer.agent

    update_events = []
    original_update = agent.update

    def timed_update():
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        result = original_update()
        end.record()
        update_events.append((start, end))
        return result

    agent.update = timed_update

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    observation, state, _ = env.reset(randomize_episode_progress=True)
    for _ in range(args.warmup):
        observation, state = trainer._rollout_and_update(observation, state)
        trainer.iteration += 1

    # ---- timed region: exactly `steps` iterations; the dominant kernel is bracketed by HIP events live
    recorder = EventRecorder()
    recorder.wrap(ops, "gather_rows", gather_bytes)
    update_events.clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        observation, state = trainer._rollout_and_update(observation, state)
        trainer.iteration += 1
    barrier()
    elapsed = time.perf_counter() - t0
    recorder.unwrap(ops)
    dominant = recorder.summary()["gather_rows"]
    update_ms = sum(s.elapsed_time(e) for s, e in update_events) / max(len(update_events), 1)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    kernels = {}
    if not args.no_kernel_pass:
        full = EventRecorder()
        for name, fn in (("buffer_push", push_bytes), ("gather_rows", gather_bytes), ("gae", gae_bytes),
                         ("next_value", next_value_bytes), ("normalize_", normalize_bytes),
                         ("ppo_loss_fwd_bwd", loss_bytes)):
            full.wrap(ops, name, fn)
        for _ in range(3):
            observation, state = trainer._rollout_and_update(observation, state)
            trainer.iteration += 1
        full.unwrap(ops)
        kernels = full.summary()

    steps_per_iteration = args.envs_per_gpu * HORIZON * world
    result = {
        "metric": "env_steps_per_sec",
        "value": round(steps_per_iteration * args.steps / elapsed, 1),
        "unit": "env-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"synthetic continuous env {args.envs_per_gpu}x obs{OBS_DIM} x act{ACT_DIM} per GPU, MLP(256,128) "
                        f"actor-critic, rollout_len={HORIZON}, ppo preset (5 epochs x 4 minibatches, Adam lr 2e-4)",
            "envs_per_gpu": args.envs_per_gpu,
            "env_steps_per_iteration": steps_per_iteration,
            "parallelism": f"dp{world}",
            "hipgraph": not args.eager,
        },
        "ppo_update_ms": round(update_ms, 3),
        "roofline": {
            "bound": "hbm",
            "kernel": "cusrl::gather_kernel (minibatch gather of all buffer leaves, one launch)",
            "achieved": dominant["achieved_GBps"],
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(dominant["achieved_GBps"] / HBM_PEAK_GBS, 4),
            "traffic": None,
            "bytes_per_launch": dominant["bytes_per_launch"],
            "avg_us": dominant["avg_us"],
            "launches": dominant["launches"],
        },
        "kernels": kernels,
    }
    trainer.environment.close()
    return result


def run_cpu_baseline(args):
    """Reference-equivalent torch CPU path on this box's host cores; bounded sample of the same workload."""
    import cusrl_amd as cusrl
    from oracle.torch_ppo import TorchPpo, run_iterations

    available = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # torch's intra-op pool does not scale to every SMT thread of a 2-socket host on these small ops (256 threads
    # measured 4 orders of magnitude slower than 8): pick the fastest pool size on a short proxy (1 epoch) iteration
    best_threads, best_time = None, float("inf")
    for threads in (8, 16, 32, 64, 128):
        if threads > available and best_threads is not None:
            break
        threads = min(threads, available)
        torch.set_num_threads(threads)
        torch.manual_seed(0)
        proxy_env = cusrl.testing.SyntheticEnvironment(NUM_ENVS, OBS_DIM, ACT_DIM, device="cpu")
        proxy = TorchPpo(OBS_DIM, ACT_DIM, NUM_ENVS, epochs=1, device="cpu")
        obs = run_iterations(proxy, proxy_env, 1)
        t0 = time.perf_counter()
        run_iterations(proxy, proxy_env, 1, obs)
        elapsed = time.perf_counter() - t0
        if elapsed < best_time:
            best_threads, best_time = threads, elapsed
        if elapsed > 1.25 * best_time or threads >= available:
            break
    cores = best_threads
    torch.set_num_threads(cores)
    torch.manual_seed(42)
    env = cusrl.testing.SyntheticEnvironment(NUM_ENVS, OBS_DIM, ACT_DIM, device="cpu")
    agent = TorchPpo(OBS_DIM, ACT_DIM, NUM_ENVS, device="cpu")
    observation = run_iterations(agent, env, 1)  # iteration 0 allocates the buffer
    iterations, t0 = 0, time.perf_counter()
    while True:
        observation = run_iterations(agent, env, 1, observation)
        iterations += 1
        elapsed = time.perf_counter() - t0
        if elapsed >= args.cpu_seconds or iterations >= 30:
            break
    return {
        "value": round(NUM_ENVS * HORIZON * iterations / elapsed, 1),
        "unit": "env-steps/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{iterations} iterations (after 1 warm-up) of the same 4096x48x12 T=24 ppo workload, "
                  f"torch {torch.__version__} CPU ops, {cores} intra-op threads (fastest of 8..128 on a proxy run; "
                  f"{available} logical CPUs visible), {elapsed:.1f} s",
        "ms_per_step": round(elapsed / iterations * 1e3, 1),
    }


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    result = run_gpu(args, rank, world)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = run_cpu_baseline(args)
            result["speedup_vs_cpu_baseline"] = round(result["value"] / result["cpu_baseline"]["value"], 2)
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
