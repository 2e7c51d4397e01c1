#!/bin/bash
# Everything profiles/r02/ is made of, in one GPU session.  Usage (on the GPU box, via gpurun): bash scripts/collect_r02.sh
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02
mkdir -p "$O/configs"
cd "$R"
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tee "$O/pytest_gpu.txt"
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > "$O/bench_line.json"
python -c "import json;d=json.load(open('$O/bench_line.json'));print('bench', d['value'], d['ms_per_step'], d['ppo_update_ms'], d['roofline']['avg_us'], d['roofline']['frac'], d.get('speedup_vs_cpu_baseline'))"
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-pass --no-scale-pass 2>/dev/null | tail -1; done | python -c "
import sys, json
for line in sys.stdin:
    d = json.loads(line); print('repeat', d['value'], d['ms_per_step'], d['ppo_update_ms'])" | tee "$O/bench_repeats.txt"
# one RCCL rank (torchrun): eager all-reduce between two graphs vs the C-ABI all-reduce captured inside the step graph
for flag in "" "--native-collectives"; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 \
      --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-pass --no-scale-pass $flag 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('rccl-1-rank', d['config']['collectives'][:48], '|', d['value'], 'env-steps/s', d['ms_per_step'], 'ms/iteration, update', d['ppo_update_ms'], 'ms')"
done | tee "$O/bench_rccl_one_rank.txt"
python scripts/kernel_bench.py --envs 4096 1048576 --json "$O/kernel_bench_graph_timed.json" 2>/dev/null | grep -v amdgpu > "$O/kernel_bench_graph_timed.txt"
tail -30 "$O/kernel_bench_graph_timed.txt"
for c in "config1" "config1 --compile" "config2 --compile" "config3 --compile" "config4" "config5" "config5 --compile"; do
  python scripts/run_config.py $c 2>&1 | grep -v amdgpu.ids > "$O/configs/run_$(echo $c | tr ' -' '__').txt"
  tail -2 "$O/configs/run_$(echo $c | tr ' -' '__').txt" | head -1 | cut -c1-160
done
bash scripts/gpu_profile.sh r02_prof --steps 20 --warmup 5 > "$O/gpu_profile.log" 2>&1
tail -45 "$O/gpu_profile.log"
bash scripts/gpu_pmc.sh r02_pmc > "$O/gpu_pmc.log" 2>&1
python - <<PY
import json
d = json.load(open("$R/gpurun_out/r02_pmc/pmc_summary.json"))
print(json.dumps(d["calibration"], indent=1))
for k in ("gather_minibatch_hot_record", "gather_minibatch_hot_leaves", "gather_minibatch_all_leaves", "gather_minibatch_all_plain", "pack_rows", "pack_hot_record"):
    if k in d: print(k, d[k]["algorithmic_bytes"], d[k]["fetch_raw"], d[k]["write_raw"], d[k]["traffic_over_algorithmic_bracket"])
PY
