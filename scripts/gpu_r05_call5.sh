#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_call5
mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -60 > "$OUT/pytest_gpu.txt"
for v in base wpe8 r2 r2wpe8 f32 f32wpe8 r1 r1wpe8 base f32wpe8; do
  echo "== variant $v"; CUSRL_HIP_LIBRARY=$R/build/variants/libcusrl_hip_$v.so timeout 200 python scripts/kernel_bench.py --envs 1048576 --only "ppo loss" 2>&1 | grep "ppo loss"
done > "$OUT/loss_variants_ab.txt" 2>&1
for s in 0 1 0 1 0 1; do
  echo "== CUSRL_STAGGER_CRITIC=$s"; CUSRL_STAGGER_CRITIC=$s timeout 200 python bench.py --no-cpu-baseline --no-kernel-pass --no-scale-pass --no-env-ab 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ppo_update_ms'])"
done > "$OUT/stagger_ab.txt" 2>&1
cat "$OUT/pytest_gpu.txt" | tail -45; cat "$OUT/loss_variants_ab.txt" "$OUT/stagger_ab.txt"
