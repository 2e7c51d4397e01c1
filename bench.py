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
    parser.add_argument("--no-timer", action="store_true", help="replace the trainer's section timer by a no-op (diagnostic)")
    parser.add_argument("--no-observer", action="store_true", help="do not bracket eager launches with HIP events")
    parser.add_argument("--autoreset", action="store_true",
                        help="env resets finished instances itself (no per-step index read-back in the trainer)")
    parser.add_argument("--no-pin", action="store_true",
                        help="leave the host thread unpinned (default: 8 CPUs of the GPU's NUMA node, cusrl_amd/utils/affinity.py)")
    return parser.parse_args()


KERNEL_NAMES = {
    "cusrl_gather_rows": "cusrl::gather_kernel",
    "cusrl_buffer_push": "cusrl::push_kernel",
    "cusrl_gae": "cusrl::gae_kernel",
    "cusrl_next_value": "cusrl::next_value_kernel",
    "cusrl_normalize": "cusrl::normalize_kernel",
    "cusrl_ppo_loss_fwd_bwd": "cusrl::ppo_loss_chunked_kernel<3> (+ finalize)",
    "cusrl_normal_sample_logp": "cusrl::normal_sample_logp_kernel",
    "cusrl_episode_stats": "cusrl::episode_stats_kernel",
}


def summarize(observer):
    """Per C-ABI entry: launches, average HIP-event time, algorithmic bytes and achieved GB/s."""
    torch.cuda.synchronize()
    out = {}
    for name, records in observer.records.items():
        if not records:
            continue
        # one entry point may be launched at several sizes (gather: the whole-buffer statistics pass and the few
        # truncated rows of the value bootstrap): report the launches of the largest size, not a mixture
        largest = max(b for _, _, b in records)
        records = [r for r in records if r[2] == largest]
        total_ms = sum(s.elapsed_time(e) for s, e, _ in records)
        total_bytes = sum(b for _, _, b in records)
        gbs = total_bytes / (total_ms * 1e-3) / 1e9 if total_ms > 0 else 0.0
        out[name] = {
            "kernel": KERNEL_NAMES.get(name, name),
            "launches": len(records),
            "avg_us": round(total_ms * 1e3 / len(records), 3),
            "bytes_per_launch": int(total_bytes / len(records)),
            "achieved_GBps": round(gbs, 1),
            "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4),
        }
    return out


def pmc_traffic(envs_per_gpu):
    """HBM bytes of the timed launch from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in separate
    passes, FETCH_SIZE doubled per the gfx950 correction — profiles/r01/pmc_summary.json, scripts/gpu_pmc.sh).
    Counters cannot be read from inside this process, so the committed measurement of the same launch is quoted."""
    path = ROOT / "profiles" / "r01" / "pmc_summary.json"
    if envs_per_gpu != NUM_ENVS or not path.exists():
        return None
    return json.loads(path.read_text())["gather_whole_buffer"]["hbm_traffic_bytes_corrected"]


def run_gpu(args, rank, world):
    import cusrl_amd as cusrl
    from cusrl_amd import ops

    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)
    pinned = []
    if not args.no_pin:
        from cusrl_amd.utils.affinity import pin_host_thread

        pinned = pin_host_thread(local_rank, cores=8, slot=local_rank)
    cusrl.config.set_device(device)
    if world > 1:
        cusrl.utils.configure_distributed()
    cusrl.set_global_seed(42)
    env = cusrl.testing.SyntheticEnvironment(args.envs_per_gpu, OBS_DIM, ACT_DIM, device=device, autoreset=args.autoreset)
    # compile=True = hipGraph replay of the act step and the minibatch steps (cusrl_amd/template/graphs.py)
    factory = cusrl.preset.PpoAgentFactory(compile=not args.eager, optimizer_kwargs={"fused": True, "capturable": True})
    trainer = cusrl.Trainer(env, factory, num_iterations=10**9, verbose=False)
    agent = trainer.agent
    if args.no_timer:
        from contextlib import nullcontext

        trainer.timer.record = lambda name, every=1: nullcontext()
        trainer.timer.__class__.__getitem__ = lambda self, name: 1.0

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

    # ---- timed region: exactly `steps` iterations.  The dominant kernel (minibatch gather) is bracketed by HIP
    # events live: 20 of its 21 launches per iteration replay inside hipGraphs (not observable from the host); the
    # 21st — the statistics pass over the whole buffer, same kernel, 2x the rows — is launched eagerly and timed.
    observer = ops.LaunchObserver(only={"cusrl_gather_rows"})
    if not args.no_observer:
        ops.set_launch_observer(observer)
    update_events.clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        observation, state = trainer._rollout_and_update(observation, state)
        trainer.iteration += 1
    barrier()
    elapsed = time.perf_counter() - t0
    ops.set_launch_observer(None)
    live = summarize(observer)
    dominant = live.get("cusrl_gather_rows", {"achieved_GBps": 0.0, "bytes_per_launch": 0, "avg_us": 0.0, "launches": 0})
    update_ms = sum(s.elapsed_time(e) for s, e in update_events) / max(len(update_events), 1)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    kernels = {}
    if not args.no_kernel_pass and rank == 0:
        # every HIP kernel of the path stand-alone at this workload's sizes, replayed from a hipGraph between one
        # HIP-event pair (bracketing each eager launch with events costs a 10-40 us stream bubble per pair and would
        # dominate these 3-10 us kernels); scripts/kernel_bench.py also reports the roofline-scale size (1M envs)
        sys.path.insert(0, str(ROOT / "scripts"))
        import kernel_bench

        for name, (us, nbytes) in kernel_bench.bench_size(args.envs_per_gpu, iters=100).items():
            kernels[name] = {"avg_us": round(us, 2), "bytes_per_launch": int(nbytes),
                             "achieved_GBps": round(nbytes / us / 1e3, 1), "frac_of_hbm_peak": round(nbytes / us / 1e3 / HBM_PEAK_GBS, 4)}

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
            "autoreset": args.autoreset,
            "host_thread_cpus": pinned,
        },
        "ppo_update_ms": round(update_ms, 3),
        "roofline": {
            "bound": "hbm",
            "kernel": "cusrl::gather_kernel (all 14 buffer leaves in one launch; timed: the eager whole-buffer launch "
                      "of the statistics pass, 98304 rows; the 20 in-graph minibatch launches move 24576 rows each)",
            "achieved": dominant["achieved_GBps"],
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(dominant["achieved_GBps"] / HBM_PEAK_GBS, 4),
            "traffic": pmc_traffic(args.envs_per_gpu),
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


def _free_port() -> int:
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def launch_ranks(args) -> int:
    """``python bench.py --gpus N`` without a launcher: start N ranks of this same script under torch.distributed.run
    (one process per GPU, rendezvous on 127.0.0.1) — the process model of cusrl/utils/config.py:31-44,160-187."""
    import subprocess

    visible = torch.cuda.device_count()
    if visible < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {visible} GPU(s) visible; refusing to report a smaller job as n_gpus={args.gpus}")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes on this driver)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(Path(__file__).resolve()), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}: the line would misreport n_gpus")
    all_cpus = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    result = run_gpu(args, rank, world)
    if all_cpus is not None:
        try:
            os.sched_setaffinity(0, all_cpus)  # the CPU baseline below gets every core back
        except OSError:
            pass
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = run_cpu_baseline(args)
            result["speedup_vs_cpu_baseline"] = round(result["value"] / result["cpu_baseline"]["value"], 2)
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
