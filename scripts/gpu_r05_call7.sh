#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_call7
mkdir -p "$OUT"
cd "$R"
timeout 600 python -m pytest tests/test_captured_step_soak.py -q -m gpu --tb=short -k "24576" 2>&1 | grep -v "^  \|warnings.warn" | cut -c1-600 | tail -15 > "$OUT/pytest_soak.txt"
for rep in 1 2; do for v in base r2f32wpe7 ntl nts ntls f32wpe7; do
  echo "== variant $v"; CUSRL_HIP_LIBRARY=$R/build/variants/libcusrl_hip_$v.so timeout 200 python scripts/kernel_bench.py --envs 1048576 --only "ppo loss" 2>&1 | grep "ppo loss"
done; done > "$OUT/loss_variants_ab.txt" 2>&1
for v in base f32 r2 f32wpe7 r2f32wpe7 base f32 r2; do
  echo "== config-2 size, variant $v"; CUSRL_HIP_LIBRARY=$R/build/variants/libcusrl_hip_$v.so timeout 200 python scripts/kernel_bench.py --envs 4096 --only "ppo loss" 2>&1 | grep "ppo loss"
done >> "$OUT/loss_variants_ab.txt" 2>&1
cat "$OUT/pytest_soak.txt" "$OUT/loss_variants_ab.txt"
