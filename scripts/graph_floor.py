#!/usr/bin/env python3
"""The device-time floor of one iteration of the bench workload: every captured region of the loop (whole rollout, pre_update
head / tail, the update, the statistics pass) replayed back to back on its own — no host between the replays, no other region —
next to the free-running iteration.  Floor vs iteration = what phase boundaries, host reads and the eager launches in between
still cost.

    python scripts/graph_floor.py [--envs 4096] [--replays 20]
"""
import argparse
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import torch  # noqa: E402

import cusrl_amd as cusrl  # noqa: E402


def timed(replay, n):
    replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--envs", type=int, default=4096)
    parser.add_argument("--replays", type=int, default=20)
    parser.add_argument("--iterations", type=int, default=40)
    args = parser.parse_args()
    device = torch.device("cuda:0")
    cusrl.config.set_device(device)
    cusrl.set_global_seed(42)
    env = cusrl.testing.SyntheticEnvironment(args.envs, 48, 12, device=device)
    factory = cusrl.preset.PpoAgentFactory(compile=True, optimizer_kwargs={"fused": True, "capturable": True})
    trainer = cusrl.Trainer(env, factory, num_iterations=10**9, verbose=False)
    agent = trainer.agent
    observation, state, _ = env.reset(randomize_episode_progress=True)

    def iterate(n):
        nonlocal observation, state
        for _ in range(n):
            observation, state = trainer._rollout_and_update(observation, state)
            trainer.iteration += 1
        trainer.flush()

    iterate(12)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iterate(args.iterations)
    torch.cuda.synchronize()
    iteration_us = (time.perf_counter() - t0) / args.iterations * 1e6

    regions = {}
    for key, entry in trainer._graphed_rollout.rollouts.items():
        regions[f"rollout {key}"] = entry["capture"]
    for hook in agent.hook:
        scratch = getattr(hook, "_replay_scratch", None)
        if scratch:
            for parity, region in scratch["heads"].items():  # (one per pinned counter, used in turn)
                regions[f"pre_update head (counter {parity})"] = region.capture
            for (bucket, parity), region in scratch["tails"].items():
                regions[f"pre_update tail (bucket {bucket}, counter {parity})"] = region.capture
        replay = getattr(hook, "_replay", None)
        if isinstance(replay, dict) and "region" in replay:
            regions["statistics pass"] = replay["region"].capture
    for key, entry in agent._graphed_epochs.epochs.items():
        regions[f"update ({key[1]} graph(s), last epoch {key[0]})"] = entry["capture"]
    total = 0.0
    seen = set()
    for name, capture in regions.items():
        if capture.graph is None:
            continue
        us = timed(capture.graph.replay, args.replays)
        nodes = capture.census.get("kernel", "?") if capture.census else "?"
        print(f"{us:9.1f} us  {name}  ({nodes} kernel nodes)")
        family = name.split(" (")[0]  # (rollout: two parities of the statistics ring; head / tail: two counters — one of each per iteration)
        if family in seen:
            continue
        seen.add(family)
        total += us
    # the eager launches between the regions: GAE + normalisation (two hooks' pre_update) and the draw of the permutations
    print(f"{total:9.1f} us  sum of the regions one iteration replays")
    print(f"{iteration_us:9.1f} us  free-running iteration ({args.iterations} iterations)")

    # where the difference sits: HIP events around every replay of a few free-running iterations (an event is a barrier packet
    # — a few microseconds each, ten per iteration), printed for the last-but-one iteration: start, duration and the gap between
    # the end of one region and the start of the next (eager launches, host reads, replay boundaries)
    from cusrl_amd.template import graphs

    names = {id(capture): name for name, capture in regions.items()}
    marks = []
    original = graphs._Capture.replay

    def replay(self):
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        original(self)
        end.record()
        marks.append((names.get(id(self), "?"), start, end))

    graphs._Capture.replay = replay
    iterate(6)
    graphs._Capture.replay = original
    torch.cuda.synchronize()
    per_iteration = len(marks) // 6
    window = marks[per_iteration * 3 : per_iteration * 5 + 1]
    origin = window[0][1]
    previous_end = None
    for name, start, end in window:
        begin, finish = origin.elapsed_time(start) * 1e3, origin.elapsed_time(end) * 1e3
        gap = "" if previous_end is None else f"gap {begin - previous_end:7.1f}"
        print(f"  t={begin:8.1f}  {finish - begin:8.1f} us  {gap:14s} {name}")
        previous_end = finish
    trainer.environment.close()


if __name__ == "__main__":
    main()
