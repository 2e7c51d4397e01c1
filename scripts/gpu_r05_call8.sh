#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_call8
mkdir -p "$OUT"
cd "$R"
for rep in 1 2; do for v in base nt_r3 nt_r3f32 nt_r2 nt_r2f32 nt_r2f32wpe7; do
  echo "== variant $v"; CUSRL_HIP_LIBRARY=$R/build/variants/libcusrl_hip_$v.so timeout 200 python scripts/kernel_bench.py --envs 1048576 --only "ppo loss" 2>&1 | grep "ppo loss"
done; done > "$OUT/loss_variants_ab.txt" 2>&1
for v in nt_r3 nt_r2f32; do
echo "== variant $v CUSRL_LOSS_WAVE_ROWS=0" >> "$OUT/loss_variants_ab.txt"
CUSRL_LOSS_WAVE_ROWS=0 CUSRL_HIP_LIBRARY=$R/build/variants/libcusrl_hip_$v.so timeout 200 python scripts/kernel_bench.py --envs 1048576 --only "ppo loss" 2>&1 | grep "ppo loss" >> "$OUT/loss_variants_ab.txt"
done
for v in nt_r3 base; do
echo "== variant $v config-2 size and 65536 envs" >> "$OUT/loss_variants_ab.txt"
CUSRL_HIP_LIBRARY=$R/build/variants/libcusrl_hip_$v.so timeout 200 python scripts/kernel_bench.py --envs 4096 65536 --only "ppo loss" 2>&1 | grep "ppo loss" >> "$OUT/loss_variants_ab.txt"
done
cat "$OUT/loss_variants_ab.txt"
