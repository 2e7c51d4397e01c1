#!/bin/bash
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -q --timeout 600 2>&1 | grep -v "tensor(\[" | tail -60 > gpurun_out/r03/test_k.log
tail -8 gpurun_out/r03/test_k.log
for i in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --no-kernel-pass --no-scale-pass 2>/dev/null | tail -1 > gpurun_out/r03/bench_k_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03/bench_k_*.json')):
    d=json.loads(open(f).read()); print(f, d['value'], d['ms_per_step'], d['ppo_update_ms'], d['roofline']['avg_us'])
PY
