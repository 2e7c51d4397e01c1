#!/bin/bash
# round 3, call A: full GPU suite + host-driven vs captured rollout A/B (interleaved, same box)
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -60 > gpurun_out/r03/test_a.log
for i in 1 2; do
  CUSRL_CAPTURE_ROLLOUT=0 timeout 300 python bench.py --no-cpu-baseline --no-kernel-pass --no-scale-pass > gpurun_out/r03/bench_host_$i.json 2> gpurun_out/r03/bench_host_$i.err
  timeout 300 python bench.py --no-cpu-baseline --no-kernel-pass --no-scale-pass > gpurun_out/r03/bench_capt_$i.json 2> gpurun_out/r03/bench_capt_$i.err
done
tail -5 gpurun_out/r03/test_a.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03/bench_*_?.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['ppo_update_ms'], d['config'].get('captured_env_steps'))
    except Exception as e: print(f, 'ERR', e)
PY
