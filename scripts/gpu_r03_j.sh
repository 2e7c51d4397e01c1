#!/bin/bash
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -q --timeout 600 2>&1 | grep -v "tensor(\[" | tail -60 > gpurun_out/r03/test_j.log
tail -8 gpurun_out/r03/test_j.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for i in 1 2; do
  CUSRL_WHOLE_ROLLOUT_GRAPH=0 timeout 300 python bench.py --no-cpu-baseline --no-kernel-pass --no-scale-pass 2>/dev/null | tail -1 > gpurun_out/r03/bench_j_steps_$i.json
  timeout 300 python bench.py --no-cpu-baseline --no-kernel-pass --no-scale-pass 2>/dev/null | tail -1 > gpurun_out/r03/bench_j_whole_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03/bench_j_*.json')):
    d=json.loads(open(f).read()); print(f, d['value'], d['ms_per_step'], d['ppo_update_ms'], d['roofline']['traffic'])
PY
