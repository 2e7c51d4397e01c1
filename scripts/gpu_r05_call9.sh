#!/bin/bash
# the rocprofv3-based parts of scripts/collect_r05.sh (they failed on a tag with a slash in its /tmp paths)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05final
mkdir -p "$O"
cd "$R"
bash scripts/gpu_pmc.sh r05final/pmc_gather > "$O/gpu_pmc.log" 2>&1
tail -3 "$O/gpu_pmc.log"
mkdir -p profiles/r05 && cp "$O/pmc_gather/pmc_summary.json" profiles/r05/pmc_gather_summary.json
python bench.py 2>/dev/null | tail -1 > "$O/bench_line.json"
python -c "import json;d=json.load(open('$O/bench_line.json'));print('bench', d['value'], d['ms_per_step'], d['ppo_update_ms'], d['roofline']['avg_us'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline'].get('frac_rocprof'), {k: v['frac'] for k, v in d['roofline']['at_scale'].items() if isinstance(v, dict)}, d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['config'].get('torch_generator_env_ms_per_step'))"
bash scripts/gpu_profile.sh r05final/prof --steps 20 --warmup 6 > "$O/gpu_profile.log" 2>&1
tail -12 "$O/gpu_profile.log"
