#!/bin/bash
# Round 6, second part: the trainer's pipelined log (CUSRL_PIPELINE_LOGS) and the compaction-first pre_update head, interleaved
# A/B on one box, plus the GPU tests that exercise the trainer loop.   Usage (through gpurun): bash scripts/gpu_r06_pipeline_ab.sh
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
timeout 900 python -m pytest tests/test_captured_rollout.py tests/test_agent_gpu.py tests/test_baseline_configs.py tests/test_observation_normalization.py tests/test_auxiliary_rewards.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -5 | tee "$O/pytest_subset.txt"
for i in 1 2 3; do
  python bench.py $B 2>/dev/null | tail -1 | brief "default (pipelined log)"
  CUSRL_PIPELINE_LOGS=0 python bench.py $B 2>/dev/null | tail -1 | brief "CUSRL_PIPELINE_LOGS=0"
done | tee "$O/pipeline_ab.txt"
python scripts/host_vs_device.py --iterations 40 2>&1 | grep -v amdgpu.ids | head -3 | cut -c1-400 | tee "$O/host_vs_device.txt"
