#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out/r04
for P in 0 1 2 3; do echo "== CUSRL_PUSH_POLICY=$P"; CUSRL_PUSH_POLICY=$P python $R/scripts/kernel_bench.py --envs 1048576 4194304 --only "push (1 step" 2>&1 | grep "push (1"; done | tee $R/gpurun_out/r04/push_policy.txt
