#!/usr/bin/env python3
"""cProfile of the host side of a few PPO iterations (config 2) — where the Python time between launches goes.

    python scripts/host_profile.py [--iterations 5] [--top 45]
"""
import argparse
import cProfile
import pstats
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import cusrl_amd as cusrl  # noqa: E402


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--envs", type=int, default=4096)
    parser.add_argument("--iterations", type=int, default=5)
    parser.add_argument("--top", type=int, default=45)
    parser.add_argument("--sort", default="tottime")
    args = parser.parse_args()
    cusrl.config.set_device("cuda:0")
    cusrl.set_global_seed(42)
    env = cusrl.testing.SyntheticEnvironment(args.envs, 48, 12, device="cuda:0")
    factory = cusrl.preset.PpoAgentFactory(compile=True, optimizer_kwargs={"capturable": True, "fused": True})
    trainer = cusrl.Trainer(env, factory, num_iterations=10**9, verbose=False)
    observation, state, _ = env.reset()
    for _ in range(4):
        observation, state = trainer._rollout_and_update(observation, state)
    torch.cuda.synchronize()
    profiler = cProfile.Profile()
    profiler.enable()
    for _ in range(args.iterations):
        observation, state = trainer._rollout_and_update(observation, state)
    torch.cuda.synchronize()
    profiler.disable()
    stats = pstats.Stats(profiler)
    stats.sort_stats(args.sort).print_stats(args.top)


if __name__ == "__main__":
    main()
