#!/usr/bin/env python3
"""The pre_update chain at roofline scale — next_value -> GAE (+ statistics) -> normalise — timed as a chain and launch by
launch, in the two cache states a kernel can meet:

  hot    the same launch back to back (what a micro-benchmark loop measures: the previous launch's lines are still in the
         256 MB Infinity Cache)
  cold   1 GiB of fresh writes in front of every timed chain (what the kernel meets in an update: the rollout's pushes and
         the previous update left the cache full of dirty lines that are not its own)

    CUSRL_GAE_POLICY=<0..7> CUSRL_GAE_BLOCK=<128|256> python scripts/pre_update_chain.py [--envs 1048576] [--json out]

The policy / block knobs exist only in builds that instantiate them (round-4 experiment, profiles/r04/gae_policy.md).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import torch  # noqa: E402

from cusrl_amd import ops  # noqa: E402

DEV = "cuda:0"


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--envs", type=int, default=1048576)
    parser.add_argument("--steps", type=int, default=24)
    parser.add_argument("--repeat", type=int, default=10)
    parser.add_argument("--json", type=str, default=None)
    args = parser.parse_args()
    T, N = args.steps, args.envs
    S = T * N
    f = lambda *shape: torch.randn(*shape, device=DEV)  # noqa: E731
    reward, value, last = f(T, N, 1), f(T, N, 1), f(N, 1)
    term = torch.rand(T, N, 1, device=DEV) < 0.01
    trunc = torch.rand(T, N, 1, device=DEV) < 0.005
    done = term | trunc
    nv, adv, ret = torch.empty_like(reward), torch.empty_like(reward), torch.empty_like(reward)
    scratch = torch.empty(1 << 28, device=DEV)  # 1 GiB

    def _event():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def chain(events=None):
        mark = (lambda: None) if events is None else (lambda: events.append(_event()))
        mark()
        ops.next_value(value, term, trunc, last, 0.0, False, nv)
        mark()
        partials = ops.gae(reward, value, nv, done, 0.99, 0.95, None, adv, ret)[2]
        mark()
        ops.normalize_from_partials_(adv, partials, S)
        mark()

    for _ in range(3):
        chain()
    torch.cuda.synchronize()
    report = {"envs": N, "steps": T, "policy": os.environ.get("CUSRL_GAE_POLICY", "default"),
              "block": os.environ.get("CUSRL_GAE_BLOCK", "default")}
    for state in ("hot", "cold"):
        sums = None
        for _ in range(args.repeat):
            if state == "cold":
                scratch.fill_(1.0)
            events: list = []
            chain(events)
            torch.cuda.synchronize()
            spans = [a.elapsed_time(b) * 1e3 for a, b in zip(events[:-1], events[1:])]
            sums = spans if sums is None else [x + y for x, y in zip(sums, spans)]
        spans = [x / args.repeat for x in sums]
        names = ["next_value", "gae", "normalize"][: len(spans)]
        algorithmic = {"next_value": 10 * S, "gae": 21 * S, "normalize": 8 * S}
        row = {n: {"us": round(us, 1), "frac": round(algorithmic[n] / us / 1e3 / 8000.0, 3)} for n, us in zip(names, spans)}
        row["chain_us"] = round(sum(spans), 1)
        row["chain_frac"] = round(sum(algorithmic[n] for n in names) / sum(spans) / 1e3 / 8000.0, 3)
        report[state] = row
    print(json.dumps(report))
    if args.json:
        with open(args.json, "a") as fh:
            fh.write(json.dumps(report) + "\n")


if __name__ == "__main__":
    main()
