#!/bin/bash
# Everything profiles/r03/ is made of, in one GPU session.  Usage (on the GPU box, via gpurun): bash scripts/collect_r03.sh
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03c
mkdir -p "$O/configs"
cd "$R"
python -m pytest tests -m gpu -q --timeout 600 2>&1 | grep -E "passed|failed|error" | tee "$O/pytest_gpu.txt"
cp gpurun_out/gradient_parity.json "$O/gradient_parity.json" 2>/dev/null
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > "$O/bench_line.json"
python -c "import json;d=json.load(open('$O/bench_line.json'));print('bench', d['value'], d['ms_per_step'], d['ppo_update_ms'], d['roofline']['avg_us'], d['roofline']['frac'], d.get('speedup_vs_cpu_baseline'), d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
brief() { python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$1', d['value'], d['ms_per_step'], d['ppo_update_ms'], d['config'].get('captured_env_steps'), d['config'].get('collectives'))"; }
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-pass --no-scale-pass 2>/dev/null | tail -1; done | brief repeat | tee "$O/bench_repeats.txt"
# A/B on this box, interleaved: the host-driven env step (fixed-shape resets, no hipGraph around the step) vs the captured one
for i in 1 2; do
  CUSRL_CAPTURE_ROLLOUT=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-pass --no-scale-pass 2>/dev/null | tail -1 | brief host_driven_step
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-pass --no-scale-pass 2>/dev/null | tail -1 | brief captured_step
done | tee "$O/bench_rollout_ab.txt"
# the loss finalize launch kept (CUSRL_DEFER_LOSS_FINALIZE=0) vs deferred (default)
for i in 1 2; do
  CUSRL_DEFER_LOSS_FINALIZE=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-pass --no-scale-pass 2>/dev/null | tail -1 | brief finalize_launch_kept
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-pass --no-scale-pass 2>/dev/null | tail -1 | brief finalize_deferred
done | tee "$O/bench_loss_finalize_ab.txt"
# one RCCL rank (torchrun): C-ABI collectives captured inside the step graph (default) vs torch.distributed's eager all-reduce
for flag in "" "--torch-collectives"; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 \
      --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-pass --no-scale-pass $flag 2>/dev/null | tail -1 | brief "rccl_one_rank$flag"
done | tee "$O/bench_rccl_one_rank.txt"
# two ranks sharing the GPU over gloo (test-only mode): launch_ranks + the multi-rank agent path + the rank-0 line
python bench.py --gpus 2 --share-gpu --steps 10 --warmup 5 --no-cpu-baseline --no-kernel-pass --no-scale-pass 2>/dev/null | grep "^{" > "$O/bench_two_ranks_share_gpu.json"
cat "$O/bench_two_ranks_share_gpu.json" | brief two_ranks_share_gpu
python scripts/kernel_bench.py --envs 4096 1048576 --json "$O/kernel_bench_graph_timed.json" 2>/dev/null | grep -v amdgpu > "$O/kernel_bench_graph_timed.txt"
tail -40 "$O/kernel_bench_graph_timed.txt"
python scripts/graph_launch_cost.py 2>&1 | grep -v amdgpu > "$O/graph_launch_cost.txt"
for c in "config1 --compile" "config2 --compile" "config3 --compile" "config4" "config5 --compile"; do
  timeout 300 python scripts/run_config.py $c --iterations 8 2>&1 | grep -v amdgpu.ids > "$O/configs/run_$(echo $c | tr ' -' '__').txt"
  tail -2 "$O/configs/run_$(echo $c | tr ' -' '__').txt" | head -1 | cut -c1-170
done
bash scripts/gpu_profile.sh r03_prof --steps 20 --warmup 6 > "$O/gpu_profile.log" 2>&1
tail -12 "$O/gpu_profile.log"
bash scripts/gpu_pmc.sh r03_pmc > "$O/gpu_pmc.log" 2>&1
tail -3 "$O/gpu_pmc.log"
