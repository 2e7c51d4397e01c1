#!/bin/bash
# Round 6 probe on one MI355X: targeted tests first, then the same-box A/B of the round's switches.  Output under gpurun_out/r06_probe/.
set -u
OUT=gpurun_out/r06_probe
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_kernels.py -q -m gpu -x -p no:cacheprovider -k "input_layer or flat_backward or fused_linear or value_term" > $OUT/pytest_kernels.txt 2>&1
tail -3 $OUT/pytest_kernels.txt
timeout 1500 python -m pytest tests/test_captured_step_soak.py -q -m gpu -x -p no:cacheprovider -k "stock or split" > $OUT/pytest_soak.txt 2>&1
tail -3 $OUT/pytest_soak.txt
B="--no-cpu-baseline --no-kernel-pass --no-scale-pass --no-env-ab"
brief() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/it, update', d['ppo_update_ms'], 'ms,', round(d['value']/1e6,2), 'M env-steps/s, epoch-graph updates', d['config'].get('epoch_graph_updates'))"; }
for i in 1 2 3; do
  for cfg in "1" "0"; do
    CUSRL_INPUT_LAYER_KERNEL=$cfg python bench.py $B --steps 50 --warmup 10 2>$OUT/bench_err_$cfg.txt | brief "input_layer_kernel=$cfg :" | tee -a $OUT/step_ab3.txt
  done
done
