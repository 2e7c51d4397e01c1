#!/usr/bin/env python3
"""Is the training loop of the bench workload host-bound?  Per iteration: the time the HOST needs to issue one rollout + update
(no synchronisation inside) next to the time the DEVICE needs to execute it, then a cProfile of the issuing thread.

    python scripts/host_vs_device.py [--iterations 40] [--profile]
"""
import argparse
import cProfile
import pstats
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import torch  # noqa: E402

import cusrl_amd as cusrl  # noqa: E402


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--iterations", type=int, default=40)
    parser.add_argument("--envs", type=int, default=4096)
    parser.add_argument("--profile", action="store_true")
    args = parser.parse_args()
    device = torch.device("cuda:0")
    cusrl.config.set_device(device)
    cusrl.set_global_seed(42)
    env = cusrl.testing.SyntheticEnvironment(args.envs, 48, 12, device=device)
    factory = cusrl.preset.PpoAgentFactory(compile=True, optimizer_kwargs={"fused": True, "capturable": True})
    trainer = cusrl.Trainer(env, factory, num_iterations=10**9, verbose=False)
    agent = trainer.agent
    observation, state, _ = env.reset(randomize_episode_progress=True)
    for _ in range(10):
        observation, state = trainer._rollout_and_update(observation, state)
        trainer.iteration += 1
    torch.cuda.synchronize()

    spans = {"update_host": 0.0, "steps_host": 0.0}
    original_update = agent.update
    from cusrl_amd.template import graphs

    original_run = graphs.GraphedTrainStep.run

    def run(self, *a, **k):
        t = time.perf_counter()
        original_run(self, *a, **k)
        spans["steps_host"] += time.perf_counter() - t

    def update():
        t = time.perf_counter()
        result = original_update()
        spans["update_host"] += time.perf_counter() - t
        return result

    graphs.GraphedTrainStep.run = run
    agent.update = update
    host = total = 0.0
    for _ in range(args.iterations):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        observation, state = trainer._rollout_and_update(observation, state)
        trainer.iteration += 1
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host += t1 - t0
        total += t2 - t0
    n = args.iterations
    print(f"per iteration, device idle at the start of each: host issue {host / n * 1e3:.3f} ms, until the device is done {total / n * 1e3:.3f} ms; "
          f"of the host time: agent.update {spans['update_host'] / n * 1e3:.3f} ms, of which the {20} GraphedTrainStep.run calls "
          f"{spans['steps_host'] / n * 1e3:.3f} ms ({spans['steps_host'] / n / 20 * 1e6:.1f} us per step)")
    # free-running (what the bench measures): no synchronisation between iterations
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iterations):
        observation, state = trainer._rollout_and_update(observation, state)
        trainer.iteration += 1
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"free-running: host returns after {(t1 - t0) / n * 1e3:.3f} ms per iteration, device done after {(t2 - t0) / n * 1e3:.3f} ms per iteration")
    if args.profile:
        profiler = cProfile.Profile()
        profiler.enable()
        for _ in range(args.iterations):
            observation, state = trainer._rollout_and_update(observation, state)
            trainer.iteration += 1
        profiler.disable()
        torch.cuda.synchronize()
        stats = pstats.Stats(profiler)
        stats.sort_stats("cumulative").print_stats(45)
        stats.sort_stats("tottime").print_stats(30)


if __name__ == "__main__":
    main()
