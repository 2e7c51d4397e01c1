#!/usr/bin/env python3
"""Synchronised wall-clock of the parts of one ``agent.update()`` (config 2): every pre_update / post_update hook, the
minibatch loop, the metric flush and the summary.  Synchronising after each part removes the overlap between them, so
the parts add up to a little more than the real update; it shows where the per-update FIXED cost sits.

    python scripts/update_phases.py
"""
import sys
import time
from collections import defaultdict
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import cusrl_amd as cusrl  # noqa: E402


def main():
    cusrl.config.set_device("cuda:0")
    cusrl.set_global_seed(42)
    env = cusrl.testing.SyntheticEnvironment(4096, 48, 12, device="cuda:0")
    factory = cusrl.preset.PpoAgentFactory(compile=True, optimizer_kwargs={"capturable": True, "fused": True})
    trainer = cusrl.Trainer(env, factory, num_iterations=10**9, verbose=False)
    agent = trainer.agent
    spent, calls = defaultdict(float), defaultdict(int)

    def timed(name, fn):
        def wrapper(*args, **kwargs):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            result = fn(*args, **kwargs)
            torch.cuda.synchronize()
            spent[name] += time.perf_counter() - t0
            calls[name] += 1
            return result
        return wrapper

    for hook in agent.hook:
        for method in ("pre_update", "post_update"):
            if getattr(type(hook), method) is not getattr(cusrl.Hook, method):
                setattr(hook, method, timed(f"{method:12s} {hook.name}", getattr(hook, method)))
    agent.sampler.iter_indices = timed("sampler.iter_indices (generator setup)", agent.sampler.iter_indices)
    base_update = cusrl.template.agent.Agent.update
    observation, state, _ = env.reset()
    for _ in range(4):
        observation, state = trainer._rollout_and_update(observation, state)
    spent.clear(), calls.clear()
    totals = []
    update = agent.update

    def outer(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = update(*a, **k)
        torch.cuda.synchronize()
        totals.append(time.perf_counter() - t0)
        return r

    agent.update = outer
    n = 10
    for _ in range(n):
        observation, state = trainer._rollout_and_update(observation, state)
    print(f"update total {sum(totals) / n * 1e3:.3f} ms (with the extra synchronisation points)")
    for name, seconds in sorted(spent.items(), key=lambda kv: -kv[1]):
        print(f"  {name:50s} {seconds / n * 1e3:7.3f} ms  ({calls[name] // n} calls)")
    print(f"  {'everything else (minibatch steps, flush, summary)':50s} {(sum(totals) - sum(spent.values())) / n * 1e3:7.3f} ms")


if __name__ == "__main__":
    main()
