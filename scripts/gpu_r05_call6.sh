#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_call6
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_captured_step_soak.py tests/test_distributed_gpu.py -q -m gpu --tb=short 2>&1 | grep -v "^  \|warnings.warn" | cut -c1-600 > "$OUT/pytest_soak_dist.txt"
for rep in 1 2; do for v in base r2 f32 r2f32 r2f32wpe7 r2f32wpe6; do
  echo "== variant $v"; CUSRL_HIP_LIBRARY=$R/build/variants/libcusrl_hip_$v.so timeout 200 python scripts/kernel_bench.py --envs 1048576 --only "ppo loss" 2>&1 | grep "ppo loss"
done; done > "$OUT/loss_variants_ab.txt" 2>&1
echo "== variant r2f32 CUSRL_LOSS_WAVE_ROWS=0" >> "$OUT/loss_variants_ab.txt"
CUSRL_LOSS_WAVE_ROWS=0 CUSRL_HIP_LIBRARY=$R/build/variants/libcusrl_hip_r2f32.so timeout 200 python scripts/kernel_bench.py --envs 4096 1048576 --only "ppo loss" 2>&1 | grep "ppo loss" >> "$OUT/loss_variants_ab.txt"
echo "== variant r2f32 config2 size" >> "$OUT/loss_variants_ab.txt"
CUSRL_HIP_LIBRARY=$R/build/variants/libcusrl_hip_r2f32.so timeout 200 python scripts/kernel_bench.py --envs 4096 --only "ppo loss" 2>&1 | grep "ppo loss" >> "$OUT/loss_variants_ab.txt"
CUSRL_HIP_LIBRARY=$R/build/variants/libcusrl_hip_base.so timeout 200 python scripts/kernel_bench.py --envs 4096 --only "ppo loss" 2>&1 | grep "ppo loss" >> "$OUT/loss_variants_ab.txt"
tail -70 "$OUT/pytest_soak_dist.txt"; cat "$OUT/loss_variants_ab.txt"
