#!/bin/bash
# Re-measure what a change to the HIP sources invalidates in profiles/r03/ (the A/B files of scripts/collect_r03.sh stay):
# GPU suite, PMC passes (bench.py quotes them only for the buffer.hip they were taken from), bench line + repeats,
# stand-alone kernel table, rocprofv3 stats of the bench command.  Usage (GPU box, via gpurun):
#   CUSRL_COMMIT=<hash> bash scripts/collect_r03_refresh.sh
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03e
mkdir -p "$O"
cd "$R"
python -m pytest tests -m gpu -q --timeout 600 2>&1 | grep -E "passed|failed|error" | tee "$O/pytest_gpu.txt"
cp gpurun_out/gradient_parity.json "$O/gradient_parity.json" 2>/dev/null
bash scripts/gpu_pmc.sh r03_pmc > "$O/gpu_pmc.log" 2>&1
tail -3 "$O/gpu_pmc.log"
cp gpurun_out/r03_pmc/pmc_summary.json profiles/r03/pmc_summary.json   # (on the box's copy: the bench line below quotes it)
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > "$O/bench_line.json"
python -c "import json;d=json.load(open('$O/bench_line.json'));print('bench', d['value'], d['ms_per_step'], d['ppo_update_ms'], d['roofline']['avg_us'], d['roofline']['frac'], d['roofline']['traffic'], d.get('speedup_vs_cpu_baseline'), d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-pass --no-scale-pass 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('repeat', d['value'], d['ms_per_step'], d['ppo_update_ms'], d['config'].get('captured_env_steps'), d['config'].get('collectives'))"; done | tee "$O/bench_repeats.txt"
python scripts/kernel_bench.py --envs 4096 1048576 --json "$O/kernel_bench_graph_timed.json" 2>/dev/null | grep -v amdgpu > "$O/kernel_bench_graph_timed.txt"
grep -i "pack\|gather hot\|loss" "$O/kernel_bench_graph_timed.txt"
bash scripts/gpu_profile.sh r03_prof --steps 20 --warmup 6 > "$O/gpu_profile.log" 2>&1
grep -A5 "^dispatches" "$O/gpu_profile.log"
