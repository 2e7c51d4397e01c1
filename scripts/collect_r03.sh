#!/bin/bash
# Everything profiles/r03/ is made of, in one GPU session.  Usage (on the GPU box, via gpurun): bash scripts/collect_r03.sh
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03d
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
# one graph replay per env step (CUSRL_WHOLE_ROLLOUT_GRAPH=0) vs one per rollout (default)
for i in 1 2; do
  CUSRL_WHOLE_ROLLOUT_GRAPH=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-pass --no-scale-pass 2>/dev/null | tail -1 | brief graph_per_env_step
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-pass --no-scale-pass 2>/dev/null | tail -1 | brief graph_per_rollout
done | tee "$O/bench_whole_rollout_ab.txt"
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
# config 5: the AMP objective through autograd's double backward vs the closed form (default); library GEMM defaults vs the
# measured selection, configs 4 and 5
for i in 1 2; do
  for v in 0 1; do
    echo "CUSRL_AMP_CLOSED_FORM=$v $(CUSRL_AMP_CLOSED_FORM=$v timeout 200 python scripts/run_config.py config5 --compile --iterations 8 2>&1 | grep iteration | tail -1 | cut -c1-60)"
  done
done | tee "$O/configs/config5_amp_closed_form_ab.txt"
for c in "config4" "config5 --compile"; do
  for v in 0 1; do
    echo "$c CUSRL_TUNED_GEMMS=$v $(CUSRL_TUNED_GEMMS=$v timeout 200 python scripts/run_config.py $c --iterations 6 2>&1 | grep iteration | tail -1 | cut -c1-60)"
  done
done | tee "$O/configs/tuned_gemms_ab.txt"
bash scripts/gpu_census_config.sh config4 r03d/census_config4 > "$O/configs/config4_kernel_census.txt" 2>&1
bash scripts/gpu_census_config.sh config5 r03d/census_config5 > "$O/configs/config5_kernel_census.txt" 2>&1
grep "per iteration" "$O/configs/config4_kernel_census.txt" "$O/configs/config5_kernel_census.txt"
bash scripts/gpu_profile.sh r03_prof --steps 20 --warmup 6 > "$O/gpu_profile.log" 2>&1
tail -12 "$O/gpu_profile.log"
bash scripts/gpu_pmc.sh r03_pmc > "$O/gpu_pmc.log" 2>&1
tail -3 "$O/gpu_pmc.log"
