#!/bin/bash
# Round 6, second part: the unjoined two-window optimizer step — the tests that hold captured steps to float64 autograd and to
# bit-identity across graph forms, then the bench A/B against the joined step (CUSRL_TWO_WINDOW_STEP=0), interleaved on one box.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06b
mkdir -p "$O"
cd "$R"
export TMPDIR=/tmp
B="--no-cpu-baseline --no-kernel-pass --no-scale-pass --no-env-ab"
brief() { python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$1', d['value'], d['ms_per_step'], d['ppo_update_ms'])"; }
timeout 1200 python -m pytest tests/test_captured_step_soak.py tests/test_agent_gpu.py tests/test_baseline_configs.py tests/test_captured_rollout.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -12 | tee "$O/pytest_two_window.txt"
for i in 1 2 3; do
  python bench.py $B 2>/dev/null | tail -1 | brief "default (two-window step)"
  CUSRL_TWO_WINDOW_STEP=0 python bench.py $B 2>/dev/null | tail -1 | brief "CUSRL_TWO_WINDOW_STEP=0"
done | tee "$O/two_window_ab.txt"
