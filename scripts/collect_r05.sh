#!/bin/bash
# Everything profiles/r05/ is made of besides the defect probes and the experiment files, in one GPU session.
# Usage (on the GPU box, via gpurun): bash scripts/collect_r05.sh
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05final
mkdir -p "$O/configs"
cd "$R"
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tee "$O/pytest_gpu.txt"
cp gpurun_out/gradient_parity.json "$O/gradient_parity.json" 2>/dev/null
# counters of the dominant kernel first: the bench line quotes them (profiles/r05/pmc_gather_summary.json) while the source matches
bash scripts/gpu_pmc.sh r05final/pmc_gather > "$O/gpu_pmc.log" 2>&1
tail -3 "$O/gpu_pmc.log"
mkdir -p profiles/r05 && cp "$O/pmc_gather/pmc_summary.json" profiles/r05/pmc_gather_summary.json  # (on the box: the line below quotes it)
python bench.py 2>/dev/null | tail -1 > "$O/bench_line.json"
python -c "import json;d=json.load(open('$O/bench_line.json'));print('bench', d['value'], d['ms_per_step'], d['ppo_update_ms'], d['roofline']['avg_us'], d['roofline']['frac'], {k: v['frac'] for k, v in d['roofline']['at_scale'].items() if isinstance(v, dict)}, d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['config'].get('torch_generator_env_ms_per_step'))"
brief() { python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$1', d['value'], d['ms_per_step'], d['ppo_update_ms'], d['config'].get('captured_env_steps'), d['config'].get('collectives'), d['config'].get('gradient_allreduce'))"; }
B="--no-cpu-baseline --no-kernel-pass --no-scale-pass --no-env-ab"
for i in 1 2 3; do python bench.py $B 2>/dev/null | tail -1; done | brief repeat | tee "$O/bench_repeats.txt"
# the driver's flags
python bench.py --gpus 1 --steps 20 --warmup 5 $B 2>/dev/null | tail -1 | brief driver_flags | tee -a "$O/bench_repeats.txt"
# interleaved A/B of the stream layout per config
for i in 1 2; do for v in 0 1; do CUSRL_CONCURRENT_CRITIC=$v python bench.py $B 2>/dev/null | tail -1 | brief "config2 CUSRL_CONCURRENT_CRITIC=$v"; done; done | tee "$O/stream_ab.txt"
for c in config1 config5; do for i in 1 2; do for v in 0 1; do
  echo "$c CUSRL_CONCURRENT_CRITIC=$v $(CUSRL_CONCURRENT_CRITIC=$v timeout 300 python scripts/run_config.py $c --compile --iterations 8 2>&1 | grep '^iteration' | tail -1 | cut -c1-60)"
done; done; done | tee -a "$O/stream_ab.txt"
# one RCCL rank (torchrun): C-ABI collectives captured inside the step graph (default), the per-network split route, torch.distributed
for flag in "" "--torch-collectives"; do for split in 0 1; do
  CUSRL_SPLIT_ALLREDUCE=$split python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 $B --time-split-route $flag 2>/dev/null | tail -1 | brief "rccl_one_rank$flag split=$split"
done; done | tee "$O/bench_rccl_one_rank.txt"
python scripts/kernel_bench.py --envs 4096 1048576 --json "$O/kernel_bench_graph_timed.json" 2>/dev/null | grep -v amdgpu > "$O/kernel_bench_graph_timed.txt"
for c in "config1 --compile" "config2 --compile" "config3 --compile" "config4" "config5 --compile"; do
  timeout 300 python scripts/run_config.py $c --iterations 8 2>&1 | grep -v amdgpu.ids > "$O/configs/run_$(echo $c | tr ' -' '__').txt"
  tail -2 "$O/configs/run_$(echo $c | tr ' -' '__').txt" | head -1 | cut -c1-170
done
for c in config1 config2 config5; do echo "== $c"; timeout 300 python scripts/graph_census.py $c 2>&1 | grep -v amdgpu.ids | tail -30; done > "$O/graph_census.txt"
bash scripts/gpu_r04_config_sequence.sh config5 r05final/configs/config5 | tail -1
bash scripts/gpu_r04_sequence.sh r05final/sequence | tail -3
bash scripts/gpu_profile.sh r05final/prof --steps 20 --warmup 6 > "$O/gpu_profile.log" 2>&1
tail -5 "$O/gpu_profile.log"
