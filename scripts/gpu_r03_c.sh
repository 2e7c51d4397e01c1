#!/bin/bash
# round 3, call C: bench after the no_grad fix, configs 1/3/5 compiled with captured rollouts, rocprofv3 of the bench
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r03/configs
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-kernel-pass --no-scale-pass 2>/dev/null | tail -1 > gpurun_out/r03/bench_c_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03/bench_c_?.json')):
    d=json.loads(open(f).read()); print(f, d['value'], d['ms_per_step'], d['ppo_update_ms'], d['config'].get('captured_env_steps'))
PY
for c in "config1 --compile" "config3 --compile" "config5 --compile"; do
  timeout 300 python scripts/run_config.py $c --iterations 8 2>&1 | grep -v amdgpu.ids > "gpurun_out/r03/configs/run_$(echo $c | tr ' -' '__').txt"
  tail -3 "gpurun_out/r03/configs/run_$(echo $c | tr ' -' '__').txt" | head -2 | cut -c1-170
done
bash scripts/gpu_profile.sh r03/prof_c --steps 20 --warmup 6 > gpurun_out/r03/gpu_profile_c.log 2>&1
tail -70 gpurun_out/r03/gpu_profile_c.log
