#!/bin/bash
# Everything profiles/r06/ is made of besides the probe / experiment files of the round, in GPU sessions of one MI355X each.
# Usage (through gpurun): bash scripts/collect_r06.sh <part>     part = tests | bench | profile | configs | third
# (third = the third part of the round: final-tree lines + one-rank pairs + rocprofv3 stats, the one-rank floors / switches /
#  pre_update timelines, the side-stream priority, step-order and noise-draw A/Bs — each script writes under gpurun_out/)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06final
mkdir -p "$O/configs"
cd "$R"
export TMPDIR=/tmp
B="--no-cpu-baseline --no-kernel-pass --no-scale-pass --no-env-ab"
brief() { python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$1', d['value'], d['ms_per_step'], d['ppo_update_ms'], d['config'].get('captured_env_steps'), d['config'].get('epoch_graph_updates'), d['config'].get('collectives'), d['config'].get('gradient_allreduce'))"; }
case "${1:-all}" in
tests)
  python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tee "$O/pytest_gpu.txt"
  cp gpurun_out/gradient_parity.json "$O/gradient_parity.json" 2>/dev/null
  ;;
bench)
  python bench.py 2>/dev/null | tail -1 > "$O/bench_line.json"
  python -c "import json;d=json.load(open('$O/bench_line.json'));print('bench', d['value'], d['ms_per_step'], d['ppo_update_ms'], d['roofline']['avg_us'], d['roofline']['frac'], d['roofline'].get('frac_graph_timed'), {k: v['frac'] for k, v in d['roofline']['at_scale'].items() if isinstance(v, dict)}, d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['config'].get('torch_generator_env_ms_per_step'))"
  for i in 1 2 3; do python bench.py $B 2>/dev/null | tail -1; done | brief repeat | tee "$O/bench_repeats.txt"
  python bench.py --gpus 1 --steps 20 --warmup 5 $B 2>/dev/null | tail -1 | brief driver_flags | tee -a "$O/bench_repeats.txt"
  python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > "$O/bench_line_driver_flags.json"
  # the same box, the round-5 tree (git worktree of the round-5 head with its own build under build/r05_tree) interleaved with
  # this one: what the round moved, free of the +-5 % between the pool's boxes
  if [ -f build/r05_tree/bench.py ]; then
    for i in 1 2 3; do
      (cd build/r05_tree && python bench.py $B --steps 50 --warmup 10 2>/dev/null | tail -1) | brief "round5_tree"
      python bench.py $B 2>/dev/null | tail -1 | brief "round6_tree"
    done | tee "$O/same_box_r05_vs_r06.txt"
  fi
  # the round's switches, one at a time against the default, interleaved
  for i in 1 2; do
    python bench.py $B 2>/dev/null | tail -1 | brief "default"
    CUSRL_PIPELINE_LOGS=0 python bench.py $B 2>/dev/null | tail -1 | brief "CUSRL_PIPELINE_LOGS=0 (log read before the next rollout is launched)"
    CUSRL_FUSED_INFERENCE=0 python bench.py $B 2>/dev/null | tail -1 | brief "CUSRL_FUSED_INFERENCE=0 (library GEMM chain for the no-grad passes)"
    CUSRL_TWO_WINDOW_STEP=0 python bench.py $B 2>/dev/null | tail -1 | brief "CUSRL_TWO_WINDOW_STEP=0 (join + one Adam launch per step)"
    CUSRL_EPOCH_GRAPHS=1 python bench.py $B 2>/dev/null | tail -1 | brief "CUSRL_EPOCH_GRAPHS=1 (one graph per epoch)"
    CUSRL_EPOCH_GRAPHS=0 python bench.py $B 2>/dev/null | tail -1 | brief "CUSRL_EPOCH_GRAPHS=0 (one graph per step)"
    CUSRL_SEPARATE_VALUE_TERM=0 python bench.py $B 2>/dev/null | tail -1 | brief "CUSRL_SEPARATE_VALUE_TERM=0"
    CUSRL_PREFETCH_GATHER=0 python bench.py $B 2>/dev/null | tail -1 | brief "CUSRL_PREFETCH_GATHER=0"
    CUSRL_INPUT_LAYER_KERNEL=0 python bench.py $B 2>/dev/null | tail -1 | brief "CUSRL_INPUT_LAYER_KERNEL=0"
    CUSRL_CONCURRENT_CRITIC=0 python bench.py $B 2>/dev/null | tail -1 | brief "CUSRL_CONCURRENT_CRITIC=0"
  done | tee "$O/switches_ab.txt"
  bash scripts/gpu_r06_one_rank.sh > "$O/one_rank.log" 2>&1; cp gpurun_out/r06_one_rank/* "$O/" 2>/dev/null; tail -6 "$O/bench_one_rank.txt"
  python scripts/input_layer_bench.py 2>&1 | grep -v amdgpu.ids | tee "$O/input_layer_bench.txt"
  python scripts/host_vs_device.py --iterations 40 2>&1 | grep -v amdgpu.ids | head -3 | cut -c1-400 | tee "$O/host_vs_device.txt"
  python scripts/graph_floor.py 2>&1 | grep -v amdgpu.ids | tee "$O/graph_floor.txt"
  python scripts/mlp_forward_bench.py 2>&1 | grep -v amdgpu.ids | tee "$O/mlp_forward_bench.txt"
  python scripts/probe_graph_fork.py 2>&1 | grep -v amdgpu.ids | tee "$O/probe_graph_fork.txt"
  ;;
profile)
  bash scripts/gpu_pmc.sh r06final/pmc_gather > "$O/gpu_pmc.log" 2>&1; tail -2 "$O/gpu_pmc.log"
  bash scripts/gpu_profile.sh r06final/prof --steps 20 --warmup 6 > "$O/gpu_profile.log" 2>&1
  tail -8 "$O/gpu_profile.log"
  bash scripts/gpu_r06_timeline.sh r06final/timeline "update 1 tail" > "$O/timeline.log" 2>&1
  python scripts/kernel_bench.py --envs 4096 1048576 --json "$O/kernel_bench_graph_timed.json" 2>/dev/null | grep -v amdgpu > "$O/kernel_bench_graph_timed.txt"
  tail -5 "$O/kernel_bench_graph_timed.txt"
  ;;
third)
  bash scripts/gpu_r06_third_part_lines.sh
  bash scripts/gpu_one_rank_floor.sh
  bash scripts/gpu_one_rank_sequence.sh
  bash scripts/gpu_side_stream_ab.sh
  bash scripts/gpu_step_order_ab.sh
  bash scripts/gpu_predraw_ab.sh
  ;;
configs)
  for c in "config1 --compile" "config2 --compile" "config3 --compile" "config4" "config5 --compile"; do
    timeout 400 python scripts/run_config.py $c --iterations 8 2>&1 | grep -v amdgpu.ids > "$O/configs/run_$(echo $c | tr ' -' '__').txt"
    tail -2 "$O/configs/run_$(echo $c | tr ' -' '__').txt" | head -1 | cut -c1-170
  done
  for c in config1 config2 config5; do echo "== $c"; timeout 300 python scripts/graph_census.py $c 2>&1 | grep -v amdgpu.ids | tail -30; done > "$O/graph_census.txt"
  tail -12 "$O/graph_census.txt"
  ;;
esac
