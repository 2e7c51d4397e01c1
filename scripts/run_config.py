#!/usr/bin/env python3
"""Run a few iterations of one of the BASELINE.json configs on cuda:0 and print timing (not the headline bench).

    python scripts/run_config.py recurrent --envs 16384 --iterations 3
    python scripts/run_config.py obsnorm   --envs 4096
"""
import argparse
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import cusrl_amd as cusrl  # noqa: E402


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("config", choices=["mlp", "recurrent", "obsnorm"])
    parser.add_argument("--envs", type=int, default=4096)
    parser.add_argument("--iterations", type=int, default=4)
    parser.add_argument("--compile", action="store_true")
    args = parser.parse_args()
    cusrl.config.set_device("cuda:0")
    cusrl.set_global_seed(42)
    env = cusrl.testing.SyntheticEnvironment(args.envs, 48, 12, device="cuda:0")
    extra = {"capturable": True, "fused": True}
    if args.config == "recurrent":
        factory = cusrl.preset.RecurrentPpoAgentFactory(rnn_type="GRU", optimizer_kwargs=extra)
    elif args.config == "obsnorm":
        factory = cusrl.preset.PpoAgentFactory(normalize_observation=True, compile=args.compile, optimizer_kwargs=extra)
    else:
        factory = cusrl.preset.PpoAgentFactory(compile=args.compile, optimizer_kwargs=extra)
    trainer = cusrl.Trainer(env, factory, num_iterations=10**9, verbose=False)
    observation, state, _ = env.reset()
    for i in range(args.iterations):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        observation, state = trainer._rollout_and_update(observation, state)
        trainer.iteration += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        info = trainer.last_info
        print(f"iteration {i}: {dt * 1e3:8.1f} ms  {args.envs * 24 / dt / 1e6:6.2f} M env-steps/s  "
              f"value_loss={info['Agent/value_loss']:.4f} kl={info['Agent/kl_divergence']:.2e} "
              f"mem={torch.cuda.max_memory_allocated() / 2**30:.2f} GiB", flush=True)


if __name__ == "__main__":
    main()
