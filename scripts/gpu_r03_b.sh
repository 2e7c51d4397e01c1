#!/bin/bash
mkdir -p gpurun_out/r03
python scripts/debug_rollout.py 2>&1 | tail -20 > gpurun_out/r03/debug_rollout.log
python -m pytest tests/test_captured_rollout.py tests/test_baseline_configs.py::test_config1_compiled_captures_the_discrete_act_step_and_trains_like_eager "tests/test_agent_gpu.py::test_critic_branch_of_the_captured_step_changes_no_bit" -m gpu -q --timeout 600 2>&1 | grep -v "^E    .*tensor(\[" | tail -150 > gpurun_out/r03/test_b.log
cat gpurun_out/r03/debug_rollout.log
