#!/usr/bin/env python3
"""Host wall-clock of the phases of one env step of the rollout loop (config 2), averaged over many steps.

    python scripts/rollout_phases.py [--steps 240]
"""
import argparse
import sys
import time
from collections import defaultdict
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import cusrl_amd as cusrl  # noqa: E402
from cusrl_amd.template.environment import update_observation_and_state  # noqa: E402


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--envs", type=int, default=4096)
    parser.add_argument("--steps", type=int, default=240)
    args = parser.parse_args()
    cusrl.config.set_device("cuda:0")
    cusrl.set_global_seed(42)
    env = cusrl.testing.SyntheticEnvironment(args.envs, 48, 12, device="cuda:0")
    factory = cusrl.preset.PpoAgentFactory(compile=True, optimizer_kwargs={"capturable": True, "fused": True})
    trainer = cusrl.Trainer(env, factory, num_iterations=10**9, verbose=False)
    agent, stats = trainer.agent, trainer.stats
    observation, state, _ = env.reset()
    for _ in range(3):
        observation, state = trainer._rollout_and_update(observation, state)
    torch.cuda.synchronize()
    spent = defaultdict(float)
    clock = time.perf_counter
    begin = clock()
    from cusrl_amd import ops

    counter = ops.HostCounter()
    slots = torch.empty(args.envs, dtype=torch.int64, device="cuda:0")
    for _ in range(args.steps):
        t0 = clock()
        action = agent.act(observation, state)
        t1 = clock()
        next_observation, next_state, reward, terminated, truncated, info = env.step(action)
        t2 = clock()
        done = torch.empty_like(terminated)
        stats.track_fused(reward, terminated, truncated, done, slots, counter.arm())  # the fused step epilogue
        t3 = clock()
        agent.inference_mode = True  # keep the buffer from filling: no update inside this loop
        agent.step(next_observation, reward, terminated, truncated, next_state, **info, done=done)
        agent.inference_mode = False
        agent.buffer.push(agent.transition)
        t4 = clock()
        indices = slots[: counter.wait()]
        t5 = clock()
        if indices.numel():
            init_observation, init_state, _ = env.reset(indices=indices)
            next_observation, next_state = trainer._splice_resets(next_observation, next_state, indices, init_observation, init_state)
        t6 = clock()
        observation, state = next_observation, next_state
        if agent.buffer.full:
            agent.buffer.clear() if hasattr(agent.buffer, "clear") else None
        for name, a, b in (("act (graph replay)", t0, t1), ("env.step", t1, t2), ("step epilogue launch", t2, t3),
                           ("agent.step + push", t3, t4), ("finished-env count (poll)", t4, t5), ("reset + patch", t5, t6)):
            spent[name] += b - a
    torch.cuda.synchronize()
    total = clock() - begin
    print(f"{args.steps} steps, {total / args.steps * 1e6:.1f} us per step")
    for name, seconds in spent.items():
        print(f"  {name:24s} {seconds / args.steps * 1e6:7.1f} us")


if __name__ == "__main__":
    main()
