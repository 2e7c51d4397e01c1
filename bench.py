#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the rollout + PPO-update loop (BASELINE.json `metric`, config 2).

One "step" = one training iteration = 24 vectorised env steps of 4096 synthetic envs (obs 48, act 12) pushed into
the HBM rollout buffer + one full `ppo`-preset update (next_value, GAE, advantage normalisation, 5 epochs x 4
minibatches of gather + fused PPO objective + backward + Adam, statistics pass).  Data are synthetic and resident
on the GPU; weights are random-init.  Multi-GPU (launched by torchrun, one rank per GPU): envs are sharded, 4096
per rank (weak scaling), gradients and advantage statistics are all-reduced over RCCL/xGMI.

Prints ONE JSON line on rank 0.  Extra objects:
  roofline       — dominant hot-path kernel = the minibatch gather that replays inside every captured train step (20 per
                   iteration): algorithmic bytes of exactly that launch / its graph-timed duration vs 8 TB/s HBM
                   roofline.at_scale — GAE, the PPO objective (std-vector form the preset runs), push, next_value,
                   normalise and the record gather at 1 048 576 envs, where a bandwidth roofline can physically be
                   approached (config 2 moves 1-13 MB per launch out of L2 / MALL)
  kernels        — the same accounting for every HIP kernel of the path at this workload's sizes
  cpu_baseline   — oracle/torch_ppo.py (reference-equivalent torch CPU path) timed on this box's host cores

`python bench.py --gpus N` without a launcher starts its own N ranks (torch.distributed.run, one process per GPU).
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
    parser.add_argument("--steps", type=int, default=50)   # >= 50 timed iterations ...
    parser.add_argument("--warmup", type=int, default=10)  # ... after >= 10 warm-up iterations
    parser.add_argument("--envs-per-gpu", type=int, default=NUM_ENVS)
    parser.add_argument("--no-cpu-baseline", action="store_true")
    parser.add_argument("--no-kernel-pass", action="store_true")
    parser.add_argument("--no-scale-pass", action="store_true", help="skip the 1 048 576-env roofline-scale kernel pass")
    parser.add_argument("--no-env-ab", action="store_true", help="skip the torch-generator-env timing beside the fused env")
    parser.add_argument("--time-split-route", action="store_true",
                        help="multi-rank runs: also time the per-network split of the gradient all-reduce (a second communicator is "
                             "created for it); implied by CUSRL_SPLIT_ALLREDUCE=1")
    parser.add_argument("--native-collectives", action="store_true",
                        help="(default since round 3, kept for old command lines) collectives through the C ABI")
    parser.add_argument("--torch-collectives", action="store_true",
                        help="force torch.distributed's collectives: eager all-reduce between two graphs per minibatch step")
    parser.add_argument("--share-gpu", action="store_true",
                        help="TEST ONLY: all ranks drive cuda:0 over a gloo process group (exercises launch_ranks, the "
                             "multi-rank agent path and the rank-0 line on a 1-GPU box; the number is not a scaling result)")
    parser.add_argument("--cpu-seconds", type=float, default=15.0)
    parser.add_argument("--eager", action="store_true", help="disable hipGraph replay (compile=False)")
    parser.add_argument("--no-timer", action="store_true", help="replace the trainer's section timer by a no-op (diagnostic)")
    parser.add_argument("--autoreset", action="store_true",
                        help="env resets finished instances itself (no per-step index read-back in the trainer)")
    parser.add_argument("--no-pin", action="store_true",
                        help="leave the host thread unpinned (default: 8 CPUs of the GPU's NUMA node, cusrl_amd/utils/affinity.py)")
    return parser.parse_args()


def pmc_traffic(envs_per_gpu):
    """HBM bytes per launch of the dominant kernel from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate
    passes, scripts/gpu_pmc.sh).  Hardware counters cannot be read from inside this process, so the committed
    measurement is QUOTED — and only while the kernel source it was taken from is the one in this tree (sha256 of
    cusrl_amd/csrc/buffer.hip recorded next to the numbers); otherwise null."""
    import hashlib

    if envs_per_gpu != NUM_ENVS:
        return None, None
    source = hashlib.sha256((ROOT / "cusrl_amd" / "csrc" / "buffer.hip").read_bytes()).hexdigest()[:16]
    for round_dir in ("r06", "r05", "r04"):
        path = ROOT / "profiles" / round_dir / "pmc_gather_summary.json"
        if not path.exists():
            continue
        summary = json.loads(path.read_text())
        entry = summary.get("gather_minibatch_hot_plain")  # the six leaves a PPO step reads, gathered plainly (round 3)
        if not entry or summary.get("buffer_hip_sha256_16") != source:
            continue
        return entry["hbm_traffic_bytes"], (f"quoted: rocprofv3 PMC passes of this kernel source ({summary.get('commit', '?')}), "
                                            f"profiles/{round_dir}/pmc_gather_summary.json; the 25 MB of sampled leaves are L2 / "
                                            "Infinity-Cache resident, the memory-side counters include the cache's hits")
    return None, None


def rocprof_gather(bytes_per_launch):
    """The dominant kernel's average duration in the committed rocprofv3 kernel trace of this very command (per-grid split of
    ``rocprofv3 --kernel-trace --stats -- python bench.py``, scripts/collect_r05.sh): the profile-side figure next to the
    graph-timed one.  The tool adds ~1.5 us to sub-10 us dispatches, so this fraction is the lower of the two.  Quoted like
    the PMC traffic: only while the kernel source is the one the profile was taken from."""
    import csv
    import hashlib

    for round_dir in ("r06", "r05", "r04"):
        path = ROOT / "profiles" / round_dir / "rocprofv3_cusrl_kernels_by_grid.csv"
        if not path.exists():
            continue
        stamp = ROOT / "profiles" / round_dir / "rocprofv3_bench_line.json"
        rows = [r for r in csv.DictReader(open(path)) if r["kernel"] == "cusrl::gather_kernel"]
        if not rows:
            continue
        row = max(rows, key=lambda r: int(r["calls"]) * int(r["grid_size_threads"]))  # the in-step minibatch gather
        us = int(row["avg_ns"]) / 1e3
        source = hashlib.sha256((ROOT / "cusrl_amd" / "csrc" / "buffer.hip").read_bytes()).hexdigest()[:16]
        recorded = None
        if stamp.exists():
            try:
                recorded = json.loads(stamp.read_text()).get("buffer_hip_sha256_16")
            except (ValueError, OSError):
                recorded = None
        if recorded is not None and recorded != source:
            continue
        return {"avg_us": round(us, 3), "calls": int(row["calls"]), "grid_size_threads": int(row["grid_size_threads"]),
                "frac": round(bytes_per_launch / us / 1e3 / HBM_PEAK_GBS, 4) if us > 0 else None,
                "source": f"profiles/{round_dir}/rocprofv3_cusrl_kernels_by_grid.csv"}
    return None


def graph_time(fn, launches=10, replays=20):
    """Average device time of ``fn`` (one launch) from a hipGraph of `launches` back-to-back calls replayed `replays`
    times between ONE HIP-event pair on the stream the graph runs on — the kernel's duration as it is inside the
    captured train step, without host launch overhead or per-launch event bubbles."""
    stream, graph = torch.cuda.Stream(), torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        fn()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=stream):
        for _ in range(launches):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        start.record()
        for _ in range(replays):
            graph.replay()
        end.record()
    torch.cuda.synchronize()
    return start.elapsed_time(end) * 1e3 / (launches * replays)  # us


def dominant_kernel(agent):
    """The minibatch gather exactly as the captured train step issues it: same buffer, same leaves (the fields the
    step's hooks read, LazyBatch), same packed record, a real permutation slice as indices."""
    from cusrl_amd import ops

    buffer = agent.buffer
    steps = list(getattr(agent, "_graphed_steps", {}).values())
    hot = sorted(steps[0].hot_fields & set(buffer.schema)) if steps else sorted(buffer.schema)
    batch = buffer.capacity * buffer.parallelism // max(len(steps), 1) if steps else buffer.capacity * buffer.parallelism // 4
    indices = torch.randperm(buffer.capacity * buffer.parallelism, device=buffer.device)[:batch].contiguous()
    buffer.prepare_sampling()
    leaves = [key for name in hot for _, key in cusrl_iterate(buffer.schema[name])]
    row_bytes = sum(ops._row_bytes(buffer.storage[key], 2) for key in leaves)
    packed = [key for key in leaves if buffer._pack is not None and key in buffer._pack.offsets]
    us = graph_time(lambda: buffer.gather(indices, fields=hot))
    nbytes = batch * (2 * row_bytes + 8)
    return {"fields": hot, "leaves": len(leaves), "packed_leaves": len(packed), "rows": batch, "row_bytes": row_bytes,
            "bytes_per_launch": nbytes, "avg_us": round(us, 3), "achieved_GBps": round(nbytes / us / 1e3, 1)}


FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X dense fp32 matrix peak: 256 CUs x 4 SIMDs x 64 flop / clk x 2.4 GHz (MI355X_MICROARCH.md)


def mlp_forward_roofline(agent):
    """``cusrl_mlp2_forward`` with the agent's own actor (48 -> 256 -> 128 -> 12) at the statistics pass's size (98 304 rows) and at
    1 048 576 rows, graph-timed like the dominant kernel: flop / duration against the fp32 MFMA peak.  The PMC side (MFMA
    instructions issued = required, matrix-pipe busy share, memory traffic) is quoted from the committed passes while the kernel
    source is the one they were taken from (profiles/r06/pmc_mlp_forward_summary.json)."""
    import hashlib

    from cusrl_amd import ops
    from cusrl_amd.nn.module import fused_inference_layers

    actor = agent.actor
    probe = torch.randn(16, actor.input_dim, device=agent.device)
    with torch.no_grad():
        layers = fused_inference_layers(actor.backbone, actor.distribution.mean_head, probe)
    if layers is None:
        return None
    w1, _, w2, _, w3, _ = layers
    per_row = 2 * (w1.numel() + w2.numel() + w3.numel())
    out = {"kernel": "cusrl::mlp2_forward_kernel (acting, value targets, statistics pass: backbone + head without autograd)",
           "bound": "mfma", "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "timing": "hipGraph of 10 launches between one HIP-event pair"}
    for rows in (98304, 1 << 20):
        x = torch.randn(rows, actor.input_dim, device=agent.device)
        with torch.no_grad():
            us = graph_time(lambda: ops.mlp2_forward(x, layers), launches=10, replays=10)
        key = "achieved" if rows == 98304 else "achieved_at_scale"
        out[key] = round(per_row * rows / us / 1e6, 1)
        out["avg_us" if rows == 98304 else "avg_us_at_scale"] = round(us, 2)
        out["frac" if rows == 98304 else "frac_at_scale"] = round(per_row * rows / us / 1e6 / FP32_MFMA_PEAK_TFLOPS, 4)
        del x
    out["rows"], out["rows_at_scale"], out["flop_per_row"] = 98304, 1 << 20, per_row
    path = ROOT / "profiles" / "r06" / "pmc_mlp_forward_summary.json"
    if path.exists():
        summary = json.loads(path.read_text())
        source = hashlib.sha256((ROOT / "cusrl_amd" / "csrc" / "mlp_forward.hip").read_bytes()).hexdigest()[:16]
        if summary.get("mlp_forward_hip_sha256_16") == source:
            out["pmc"] = {name: {k: entry.get(k) for k in ("mfma_issued_over_required", "mfma_pipe_utilisation", "traffic_over_algorithmic")}
                          for name, entry in summary["cases"].items() if name != "stream_16B"}
            out["pmc_source"] = "quoted: rocprofv3 PMC passes of this kernel source, profiles/r06/pmc_mlp_forward_summary.json"
    return out


def cusrl_iterate(schema):
    from cusrl_amd.utils.nest import iterate_nested

    return iterate_nested(schema)


def torch_generator_env_ms(args, device):
    """ms per iteration of the same workload with the synthetic env in its torch-generator form (``fused=False``): the A/B that
    separates what the benchmark's env fixture contributes to the headline from what the path does."""
    import cusrl_amd as cusrl

    cusrl.set_global_seed(42)
    env = cusrl.testing.SyntheticEnvironment(args.envs_per_gpu, OBS_DIM, ACT_DIM, device=device, autoreset=args.autoreset, fused=False)
    factory = cusrl.preset.PpoAgentFactory(compile=not args.eager, optimizer_kwargs={"fused": True, "capturable": True})
    trainer = cusrl.Trainer(env, factory, num_iterations=10**9, verbose=False)
    observation, state, _ = env.reset(randomize_episode_progress=True)
    for _ in range(max(args.warmup, 6)):  # (every graph of the loop is captured by iteration 4)
        observation, state = trainer._rollout_and_update(observation, state)
        trainer.iteration += 1
    steps = min(args.steps, 20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        observation, state = trainer._rollout_and_update(observation, state)
        trainer.iteration += 1
    trainer.flush()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    env.close()
    del trainer
    torch.cuda.empty_cache()
    return round(ms, 3)


def run_gpu(args, rank, world):
    import cusrl_amd as cusrl
    from cusrl_amd import ops

    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    device = torch.device(f"cuda:{0 if args.share_gpu else local_rank}")
    torch.cuda.set_device(device)
    pinned = []
    if not args.no_pin:
        from cusrl_amd.utils.affinity import pin_host_thread

        pinned = pin_host_thread(0 if args.share_gpu else local_rank, cores=8, slot=local_rank)
    cusrl.config.set_device(device)
    cusrl.config.native_collectives = not args.torch_collectives
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

    # ---- timed region: exactly `steps` iterations, bracketed by barrier + synchronize on both sides
    update_events.clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        observation, state = trainer._rollout_and_update(observation, state)
        trainer.iteration += 1
    trainer.flush()  # the last iteration's log (the trainer reads an iteration's log behind the next rollout's launch) is timed too
    barrier()
    elapsed = time.perf_counter() - t0
    update_ms = sum(s.elapsed_time(e) for s, e in update_events) / max(len(update_events), 1)
    # the dominant kernel replays inside hipGraphs (20 launches per iteration) where it cannot be bracketed from the
    # host: time the identical launch stand-alone from a graph right after the timed region
    dominant = dominant_kernel(agent) if rank == 0 else None
    # the gradient all-reduce of one optimizer step at this run's world size (a collective: EVERY rank times it), as the
    # step issues it: the C-ABI call captured in a hipGraph, or torch.distributed's eager call between two graphs
    allreduce_us = split_allreduce_us = None
    if torch.distributed.is_available() and torch.distributed.is_initialized() and agent.flat_gradients is not None:
        flat = agent.flat_gradients.buffer
        scratch = torch.zeros_like(flat)
        comm = cusrl.utils.distributed.native_comm()
        if comm is not None:
            allreduce_us = graph_time(lambda: comm.allreduce_mean_(scratch))
            # ... and the per-network split route (CONFIG.split_gradient_allreduce): the critic's window on a side stream through
            # a second communicator next to the rest on the main stream — both routes' durations in every multi-rank line, so
            # that the first 8-GPU session is one A/B (the split route's point is the overlap with the actor's backward, which
            # this stand-alone figure does not contain: it says what the two half-size collectives cost side by side)
            # (only when asked — `--time-split-route`, or the route is switched on: two communicators with collectives in flight
            # on two streams is exactly the hazard the route carries, and the default scaling run must not depend on it)
            second = cusrl.utils.distributed.branch_comm()
            if second is None and args.time_split_route:
                second, _ = cusrl.utils.distributed.establish_native_comm(cusrl.utils.distributed.RcclComm.from_process_group, device,
                                                                         rank, world)
            if second is not None:
                critic = {id(p) for p in agent.critic.parameters()}
                ids = [i for i, p in enumerate(agent.flat_gradients.params) if id(p) in critic]
                lo = agent.flat_gradients.offsets[ids[0]]
                hi = agent.flat_gradients.offsets[ids[-1] + 1] if ids[-1] + 1 < len(agent.flat_gradients.offsets) else flat.numel()
                side = torch.cuda.Stream(device=device)

                def split_route():
                    main = torch.cuda.current_stream()
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        second.allreduce_mean_(scratch[lo:hi])
                    if lo > 0:
                        comm.allreduce_mean_(scratch[:lo])
                    if hi < scratch.numel():
                        comm.allreduce_mean_(scratch[hi:])
                    main.wait_stream(side)

                try:  # (every rank runs the same code on the same stack: a failure here is a failure on all of them)
                    split_allreduce_us = graph_time(split_route)
                except Exception as error:
                    print(f"bench: split-route all-reduce timing skipped ({type(error).__name__}: {error})", file=sys.stderr)
        else:
            torch.cuda.synchronize()
            t_ar = time.perf_counter()
            for _ in range(50):
                torch.distributed.all_reduce(scratch)
            torch.cuda.synchronize()
            allreduce_us = (time.perf_counter() - t_ar) * 1e6 / 50

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    env_ab_ms = None
    if getattr(env, "fused", False) and world == 1 and not args.no_env_ab:
        try:
            env_ab_ms = torch_generator_env_ms(args, device)
        except Exception as error:  # an A/B beside the headline must never cost the line itself
            print(f"bench: torch-generator env A/B skipped ({type(error).__name__}: {error})", file=sys.stderr)

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

    scale = {}
    if not args.no_scale_pass and rank == 0 and world == 1:
        # the north-star bar "GAE + loss kernels >= 40 % of the HBM roofline" needs sizes where a launch outlives its
        # latency: the same kernels at 1 048 576 envs (24 M slots), graph-timed by scripts/kernel_bench.py
        sys.path.insert(0, str(ROOT / "scripts"))
        import kernel_bench

        wanted = {"gae + return + stats": "gae", "ppo loss fwd+bwd, std vector": "ppo_loss_std_vector",
                  "ppo loss fwd+bwd (": "ppo_loss_std_matrix", "push (1 step": "push", "next_value": "next_value",
                  "normalize": "normalize", "gather hot leaves from the 256 B hot record": "gather_hot_record"}
        for name, (us, nbytes) in kernel_bench.bench_size(1 << 20, only=("gae + return", "ppo loss", "push (1 step", "next_value",
                                                                        "normalize", "gather hot leaves from"), iters=10).items():
            for prefix, key in wanted.items():
                if name.startswith(prefix):
                    scale[key] = {"avg_us": round(us, 1), "bytes_per_launch": int(nbytes),
                                  "achieved_GBps": round(nbytes / us / 1e3, 1),
                                  "frac": round(nbytes / us / 1e3 / HBM_PEAK_GBS, 4)}
        torch.cuda.empty_cache()

    mfma = None
    if not args.no_scale_pass and rank == 0 and world == 1:
        try:
            mfma = mlp_forward_roofline(agent)
        except Exception as error:  # an extra beside the headline must never cost the line itself
            print(f"bench: cusrl_mlp2_forward roofline skipped ({type(error).__name__}: {error})", file=sys.stderr)
        torch.cuda.empty_cache()

    steps_per_iteration = args.envs_per_gpu * HORIZON * world
    traffic, traffic_source = pmc_traffic(args.envs_per_gpu)
    profile = rocprof_gather(dominant["bytes_per_launch"]) if dominant and args.envs_per_gpu == NUM_ENVS else None
    dominant = dominant or {"achieved_GBps": 0.0, "bytes_per_launch": 0, "avg_us": 0.0, "rows": 0, "fields": [], "leaves": 0,
                            "packed_leaves": 0, "row_bytes": 0}
    # a one-rank torchrun job still runs every collective (process group of one): report it as what it is
    in_group = torch.distributed.is_available() and torch.distributed.is_initialized()
    backend = torch.distributed.get_backend() if in_group else None
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
            "backend": "rccl (torch.distributed nccl)" if backend == "nccl" else backend,
            "rccl_ranks": world if backend == "nccl" else 0,
            "collectives": cusrl.utils.distributed.collective_route(),
            "gradient_allreduce": None if allreduce_us is None else {
                "floats": int(agent.flat_gradients.buffer.numel()), "avg_us": round(allreduce_us, 2), "per_iteration": 20,
                "split_route_avg_us": None if split_allreduce_us is None else round(split_allreduce_us, 2),
                "split_route_active": bool(getattr(agent, "_split_plan", None)),
                "timing": "hipGraph of 10 calls x 20 replays" if cusrl.utils.distributed.native_comm() is not None
                          else "50 eager calls, host clock"},
            **({"share_gpu": True, "test_only": "all ranks drive cuda:0 over gloo: exercises the multi-rank path on one GPU, "
                                                "NOT a scaling measurement"} if args.share_gpu else {}),
            "hipgraph": not args.eager,
            # the benchmark's synthetic env steps as ONE launch (cusrl_synthetic_env_step); the torch-generator form of the same
            # env (5 generator launches per step) is timed beside it below: the difference is the benchmark fixture, not the path
            "fused_env": bool(getattr(env, "fused", False)),
            "torch_generator_env_ms_per_step": env_ab_ms,
            "captured_env_steps": (trainer._graphed_rollout.captured if trainer._graphed_rollout is not None else 0),
            "epoch_graph_updates": (agent._graphed_epochs.replays if getattr(agent, "_graphed_epochs", None) is not None else 0),
            # minibatch steps that left their two streams unjoined (per-network assembly + one step launch per window); with several
            # ranks: ONE all-reduce on the main stream behind both assemblies, cusrl_adam_step_normed per window
            "unjoined_steps_captured": int(getattr(getattr(agent, "flat_optimizer", None), "two_window_steps", 0)),
            "autoreset": args.autoreset,
            "host_thread_cpus": pinned,
        },
        "ppo_update_ms": round(update_ms, 3),
        "roofline": {
            "bound": "hbm",
            "kernel": f"cusrl::gather_kernel — the minibatch gather (20 launches per iteration; since round 6 issued one step "
                      f"AHEAD, at the tail of the previous step's critic branch inside the epoch's hipGraph): "
                      f"{dominant['rows']} sampled slots x {dominant['leaves']} leaves the step reads ({', '.join(dominant['fields'])}; "
                      f"{dominant['packed_leaves']} of them through the per-slot record — none while the sampled leaves fit L2 + "
                      f"Infinity Cache, Buffer.record_threshold_bytes), {dominant['row_bytes']} B/slot read + written + 8 B index",
            # `frac` / `achieved` are the REPRODUCIBLE figures: algorithmic bytes of the launch / its average duration in the
            # committed rocprofv3 kernel trace of this very command (per-grid split, profiles/<round>/) / 8 TB/s.  The same
            # launch timed live by this run (a hipGraph of 10 identical launches x 20 replays between one HIP-event pair,
            # right after the timed region, on the graph's stream — the in-step launch cannot be bracketed from the host) is
            # beside it as `frac_graph_timed`: the profiler adds ~1.5-2 us to a sub-10 us dispatch and the in-step launch shares
            # the chip with the other branch, so the live figure is the higher of the two.  Without a committed profile of this
            # kernel source `frac` falls back to the live figure and `frac_source` says so.
            "timing": ("rocprofv3 --kernel-trace per-grid average of this command: " + profile["source"] if profile
                       else "graph-timed live (no committed rocprofv3 profile matches this kernel source)"),
            "frac_source": "rocprofv3" if profile else "graph_timed",
            "achieved": (round(dominant["bytes_per_launch"] / profile["avg_us"] / 1e3, 1) if profile else dominant["achieved_GBps"]),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": (profile["frac"] if profile else round(dominant["achieved_GBps"] / HBM_PEAK_GBS, 4)),
            "frac_graph_timed": round(dominant["achieved_GBps"] / HBM_PEAK_GBS, 4),
            "achieved_graph_timed": dominant["achieved_GBps"],
            "frac_rocprof": (profile or {}).get("frac"),
            "rocprof": profile,
            "traffic": traffic,
            "traffic_source": traffic_source,
            "bytes_per_launch": dominant["bytes_per_launch"],
            "avg_us": (profile["avg_us"] if profile else dominant["avg_us"]),
            "avg_us_graph_timed": dominant["avg_us"],
            "launches_per_iteration": 20,
            "note": "GAE stages its (reward, value, next_value, done) tuple in registers, not LDS: every element is used "
                    "once by the lane that loaded it (DESIGN.md section 3)",
            # the north-star criterion (GAE + loss >= 40 % of the HBM roofline) where a roofline can physically be shown:
            # the same C-ABI launches at 1 048 576 envs x 24 steps, algorithmic bytes / graph-timed duration / 8 TB/s
            "at_scale": {"envs": 1 << 20, "timing": "hipGraph of 10 launches between one HIP-event pair (scripts/kernel_bench.py)",
                         "counters": "profiles/r04/pmc/pmc_summary.json, profiles/r05/pmc/pmc_summary.json (kernel sources unchanged since)",
                         **scale},
            # ... and flat, for consumers that keep scalars only
            **{f"at_scale_{key}_frac": entry["frac"] for key, entry in scale.items()},
            # the one kernel of the path bound by the matrix cores, not by bytes (round 6): the no-grad MLP pass
            **({"mfma_kernel": mfma} if mfma else {}),
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
    if visible < args.gpus and not args.share_gpu:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {visible} GPU(s) visible; refusing to report a smaller job as n_gpus={args.gpus}")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes on this driver)
    env.setdefault("OMP_NUM_THREADS", "8")
    if args.share_gpu:
        env["CUSRL_SHARE_GPU"] = "1"
    for attempt in range(3):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(Path(__file__).resolve()), *sys.argv[1:]]
        done = subprocess.run(cmd, env=env, stderr=subprocess.PIPE, text=True)
        sys.stderr.write(done.stderr)
        # the port was free when it was picked; somebody (an outgoing connection's source port) may have taken it since:
        # that, and only that, is retried — before any rank has done work
        if done.returncode == 0 or "EADDRINUSE" not in done.stderr:
            return done.returncode
    return done.returncode


def main():
    args = parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))
    if args.share_gpu:
        os.environ["CUSRL_SHARE_GPU"] = "1"  # before cusrl_amd reads its process configuration
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
    if world > 1 and torch.distributed.is_available() and torch.distributed.is_initialized():
        # shutdown of a multi-rank job: the line is out and every rank is past the last collective.  Neither interpreter exit
        # nor destroy_process_group() is a safe way out — both race ProcessGroupNCCL's watchdog thread, which polls its works'
        # events while communicator and events are being destroyed (a sporadic abort at exit would fail the whole torchrun
        # job after the measurement) — so the process ends its threads with itself
        torch.cuda.synchronize()
        torch.distributed.barrier()
        torch.cuda.synchronize()
        sys.stdout.flush(), sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
