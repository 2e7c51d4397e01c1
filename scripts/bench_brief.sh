#!/bin/bash
# N short bench runs, one line each: ms per iteration, ms in agent.update, env-steps/s.  Usage: scripts/bench_brief.sh [N] [bench args]
N=${1:-3}; shift || true
for i in $(seq 1 "$N"); do
  timeout 300 python bench.py --no-cpu-baseline --no-kernel-pass --no-scale-pass "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print(f\"iter {d['ms_per_step']:.3f} ms  update {d['ppo_update_ms']:.3f} ms  rollout {d['ms_per_step'] - d['ppo_update_ms']:.3f} ms  {d['value'] / 1e6:.3f} M env-steps/s\")"
done
