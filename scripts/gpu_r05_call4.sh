#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_call4
mkdir -p "$OUT"
cd "$R"
for t in 1 4; do
  echo "== parity tests, CUSRL_LOSS_TILES=$t"; CUSRL_LOSS_TILES=$t timeout 300 python -m pytest tests/test_hip_kernels.py tests/test_oracle_golden.py -q -m gpu -k "loss or objective or ppo" 2>&1 | tail -4
done > "$OUT/loss_parity.txt" 2>&1
for t in 1 2 4 8 1 4; do
  echo "== CUSRL_LOSS_TILES=$t"; CUSRL_LOSS_TILES=$t timeout 200 python scripts/kernel_bench.py --envs 1048576 --only "ppo loss" 2>&1 | grep "ppo loss"
done > "$OUT/loss_tiles_ab.txt" 2>&1
cat "$OUT/loss_parity.txt" "$OUT/loss_tiles_ab.txt"
