#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05final6
mkdir -p "$O"
cd "$R"
python bench.py 2>/dev/null | tail -1 > "$O/bench_line.json"
python -c "import json;d=json.load(open('$O/bench_line.json'));print('bench', d['value'], d['ms_per_step'], d['ppo_update_ms'], d['roofline']['frac'], d['roofline'].get('frac_rocprof'), {k: v['frac'] for k, v in d['roofline']['at_scale'].items() if isinstance(v, dict)}, d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['config'].get('torch_generator_env_ms_per_step'))"
bash scripts/gpu_profile.sh r05final6/prof --steps 20 --warmup 6 > "$O/gpu_profile.log" 2>&1
tail -7 "$O/gpu_profile.log"
bash scripts/gpu_r04_sequence.sh r05final6/sequence | tail -2
