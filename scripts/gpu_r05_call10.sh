#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_call10
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_auxiliary_rewards.py tests/test_baseline_configs.py tests/test_captured_rollout.py tests/test_captured_step_soak.py tests/test_agent_gpu.py -q -m gpu --tb=short -k "amp or config5 or rnd or aux or Amp or AMP" 2>&1 | grep -v "^  \|warnings.warn" | cut -c1-500 | tail -25 > "$OUT/pytest_amp.txt"
for i in 1 2; do timeout 300 python scripts/run_config.py config5 --compile --iterations 8 2>&1 | grep '^iteration' | tail -2; done > "$OUT/config5.txt"
timeout 300 python scripts/graph_census.py config5 2>&1 | grep -v amdgpu.ids | tail -14 | cut -c1-220 > "$OUT/census5.txt"
cat "$OUT/pytest_amp.txt" "$OUT/config5.txt" "$OUT/census5.txt"
