#!/bin/bash
set -u
OUT=gpurun_out/r06_probe
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_captured_step_soak.py tests/test_agent_gpu.py tests/test_baseline_configs.py -q -m gpu -x -p no:cacheprovider -k "not float64" > $OUT/pytest_sel.txt 2>&1
tail -3 $OUT/pytest_sel.txt
B="--no-cpu-baseline --no-kernel-pass --no-scale-pass --no-env-ab"
brief() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/it, update', d['ppo_update_ms'], 'ms,', round(d['value']/1e6,2), 'M env-steps/s')"; }
for i in 1 2 3; do
  for mode in update 1; do
    CUSRL_EPOCH_GRAPHS=$mode python bench.py $B 2>$OUT/bench_err.txt | brief "CUSRL_EPOCH_GRAPHS=$mode :" | tee -a $OUT/step_ab6.txt
  done
done
