#!/bin/bash
# Everything profiles/r04/ is made of (besides the experiment files of the GAE / push cache-policy work), in one GPU session.
# Usage (on the GPU box, via gpurun): bash scripts/collect_r04.sh
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04final
mkdir -p "$O/configs"
cd "$R"
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tee "$O/pytest_gpu.txt"
cp gpurun_out/gradient_parity.json "$O/gradient_parity.json" 2>/dev/null
python bench.py 2>/dev/null | tail -1 > "$O/bench_line.json"
python -c "import json;d=json.load(open('$O/bench_line.json'));print('bench', d['value'], d['ms_per_step'], d['ppo_update_ms'], d['roofline']['avg_us'], d['roofline']['frac'], {k: v['frac'] for k, v in d['roofline']['at_scale'].items() if isinstance(v, dict)}, d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
brief() { python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$1', d['value'], d['ms_per_step'], d['ppo_update_ms'], d['config'].get('captured_env_steps'), d['config'].get('epoch_graph_updates'), d['config'].get('collectives'), d['config'].get('gradient_allreduce'))"; }
B="--no-cpu-baseline --no-kernel-pass --no-scale-pass"
for i in 1 2 3; do python bench.py $B 2>/dev/null | tail -1; done | brief repeat | tee "$O/bench_repeats.txt"
# interleaved A/B on this box of this round's switches
for i in 1 2; do for v in 0 1; do CUSRL_FUSED_ENV=$v python bench.py $B --steps 80 --warmup 10 2>/dev/null | tail -1 | brief "CUSRL_FUSED_ENV=$v"; done; done | tee "$O/bench_fused_env_ab.txt"
for i in 1 2; do for v in 0 1; do CUSRL_EPOCH_GRAPHS=$v python bench.py $B --steps 80 --warmup 10 2>/dev/null | tail -1 | brief "CUSRL_EPOCH_GRAPHS=$v"; done; done | tee "$O/bench_epoch_graphs_ab.txt"
for i in 1 2; do for v in 0 1; do CUSRL_FUSE_EPILOGUE_PUSH=$v python bench.py $B --steps 80 --warmup 10 2>/dev/null | tail -1 | brief "CUSRL_FUSE_EPILOGUE_PUSH=$v"; done; done | tee "$O/bench_epilogue_push_ab.txt"
# one RCCL rank (torchrun): C-ABI collectives captured inside the step graph (default) vs torch.distributed's eager all-reduce
for flag in "" "--torch-collectives"; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 $B $flag 2>/dev/null | tail -1 | brief "rccl_one_rank$flag"
done | tee "$O/bench_rccl_one_rank.txt"
python scripts/kernel_bench.py --envs 4096 1048576 --json "$O/kernel_bench_graph_timed.json" 2>/dev/null | grep -v amdgpu > "$O/kernel_bench_graph_timed.txt"
for N in 1048576 4194304; do python scripts/pre_update_chain.py --envs $N; CUSRL_GAE_POLICY=0 CUSRL_GAE_BLOCK=256 python scripts/pre_update_chain.py --envs $N; done 2>/dev/null | grep "^{" > "$O/pre_update_chain.jsonl"
for c in "config1 --compile" "config2 --compile" "config3 --compile" "config4" "config5 --compile"; do
  timeout 300 python scripts/run_config.py $c --iterations 8 2>&1 | grep -v amdgpu.ids > "$O/configs/run_$(echo $c | tr ' -' '__').txt"
  tail -2 "$O/configs/run_$(echo $c | tr ' -' '__').txt" | head -1 | cut -c1-170
done
bash scripts/gpu_r04_config_sequence.sh config4 r04final/configs/config4 gru_gates_fwd ppo_loss_rowgroup | tail -1
bash scripts/gpu_r04_config_sequence.sh config5 r04final/configs/config5 | tail -1
bash scripts/gpu_r04_sequence.sh r04final/sequence | tail -3
bash scripts/gpu_profile.sh r04final/prof --steps 20 --warmup 6 > "$O/gpu_profile.log" 2>&1
tail -5 "$O/gpu_profile.log"
bash scripts/gpu_pmc.sh r04final/pmc_gather > "$O/gpu_pmc.log" 2>&1
tail -3 "$O/gpu_pmc.log"
bash scripts/gpu_pmc_r04.sh r04final/pmc > "$O/gpu_pmc_r04.log" 2>&1
tail -3 "$O/gpu_pmc_r04.log"
