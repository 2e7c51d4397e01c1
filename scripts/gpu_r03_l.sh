#!/bin/bash
mkdir -p gpurun_out/r03
for i in 1 2 3; do
  CUSRL_INPLACE_INDICES=0 timeout 300 python bench.py --no-cpu-baseline --no-kernel-pass --no-scale-pass 2>/dev/null | tail -1 > gpurun_out/r03/bench_l_copy_$i.json
  timeout 300 python bench.py --no-cpu-baseline --no-kernel-pass --no-scale-pass 2>/dev/null | tail -1 > gpurun_out/r03/bench_l_inplace_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03/bench_l_*.json')):
    d=json.loads(open(f).read()); print(f, d['value'], d['ms_per_step'], d['ppo_update_ms'], d['roofline']['avg_us'])
PY
