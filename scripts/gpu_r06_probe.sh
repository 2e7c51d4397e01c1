#!/bin/bash
# Round 6 probe on one MI355X: targeted tests, idle-gap census, bench repeats.  Output under gpurun_out/r06_probe/.
set -u
OUT=gpurun_out/r06_probe
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_agent_gpu.py tests/test_captured_step_soak.py tests/test_captured_rollout.py tests/test_baseline_configs.py -q -m gpu -x -p no:cacheprovider > $OUT/pytest_sel.txt 2>&1
tail -4 $OUT/pytest_sel.txt
B="--no-cpu-baseline --no-kernel-pass --no-scale-pass --no-env-ab"
brief() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/it, update', d['ppo_update_ms'], 'ms,', round(d['value']/1e6,2), 'M env-steps/s, epoch-graph updates', d['config'].get('epoch_graph_updates'))"; }
for i in 1 2 3; do
  python bench.py $B --steps 50 --warmup 10 2>$OUT/bench_err.txt | brief "default :" | tee -a $OUT/step_ab4.txt
done
