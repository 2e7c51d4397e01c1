#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
for ROWS in 4 8 16 32; do echo "== CUSRL_GRU_BIAS_ROWS=$ROWS"; CUSRL_GRU_BIAS_ROWS=$ROWS python $R/scripts/run_config.py config4 --iterations 7 2>&1 | grep "iteration [456]"; done
