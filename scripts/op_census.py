#!/usr/bin/env python3
"""Count the aten ops (~ kernel launches) one PPO iteration dispatches, split into rollout and update.

The hipGraph path replays whatever the eager path launches, so an eager census (``compile=False``) tells which
torch ops are still in the per-minibatch step and are candidates for folding into a cusrl_* kernel.

    python scripts/op_census.py --envs 4096 [--top 40]
"""
import argparse
import sys
from collections import Counter
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

import cusrl_amd as cusrl  # noqa: E402

VIEW_OPS = ("view", "reshape", "expand", "slice", "select", "unsqueeze", "squeeze", "t.", "transpose", "detach",
            "alias", "as_strided", "unbind", "split", "permute", "_unsafe_view", "unflatten", "flatten", "chunk",
            "narrow", "lift_fresh", "empty", "is_", "size", "stride", "numel", "item", "_local_scalar")


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.counts = Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).removeprefix("aten.")
        if not name.startswith(VIEW_OPS):
            shapes = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)][:2]
            self.counts[f"{name} {shapes}"] += 1
        return func(*args, **(kwargs or {}))


def report(title, counts, top):
    total = sum(counts.values())
    print(f"== {title}: {total} non-view aten ops")
    for name, n in counts.most_common(top):
        print(f"{n:6d}  {name}")


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--envs", type=int, default=4096)
    parser.add_argument("--top", type=int, default=60)
    parser.add_argument("--config", default=None, help="a BASELINE config of scripts/run_config.py instead of the ppo preset")
    args = parser.parse_args()
    cusrl.config.set_device("cuda:0")
    cusrl.set_global_seed(42)
    if args.config:
        sys.path.insert(0, str(Path(__file__).resolve().parent))
        import run_config

        env, factory = run_config.build(args.config, None, False)
    else:
        env = cusrl.testing.SyntheticEnvironment(args.envs, 48, 12, device="cuda:0")
        factory = cusrl.preset.PpoAgentFactory(optimizer_kwargs={"capturable": True, "fused": True})
    trainer = cusrl.Trainer(env, factory, num_iterations=10**9, verbose=False)
    observation, state, _ = env.reset()
    for _ in range(2):
        observation, state = trainer._rollout_and_update(observation, state)
    agent = trainer.agent
    update = agent.update
    rollout_census, update_census = Census(), Census()

    def counted_update(*a, **kw):
        rollout_census.__exit__(None, None, None)
        with update_census:
            result = update(*a, **kw)
        rollout_census.__enter__()
        return result

    agent.update = counted_update
    with rollout_census:
        trainer._rollout_and_update(observation, state)
    torch.cuda.synchronize()
    report("rollout (24 steps)", rollout_census.counts, args.top)
    report("update (20 minibatches)", update_census.counts, args.top)


if __name__ == "__main__":
    main()
