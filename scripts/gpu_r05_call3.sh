#!/bin/bash
# Round 5, third GPU call: the new tests + whole GPU suite, the full probe (committed evidence), loss-layout A/B + PMC.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_call3
mkdir -p "$OUT"
cd "$R"
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -40 > "$OUT/pytest_gpu.txt"
for w in 0 1 0 1; do
  echo "== CUSRL_LOSS_WAVE_ROWS=$w"; CUSRL_LOSS_WAVE_ROWS=$w timeout 200 python scripts/kernel_bench.py --envs 4096 1048576 --only "ppo loss" 2>&1 | grep -v amdgpu.ids | tail -6
done > "$OUT/loss_layout_ab.txt" 2>&1
bash scripts/gpu_pmc_r05.sh r05_call3/pmc > "$OUT/pmc_log.txt" 2>&1
timeout 500 python scripts/probe_aten_reduce_capture.py 2000 2>&1 | grep -v amdgpu.ids > "$OUT/aten_reduce_probe.txt"
cat "$OUT/pytest_gpu.txt"; cat "$OUT/loss_layout_ab.txt"; tail -15 "$OUT/pmc_log.txt"; cut -c1-300 "$OUT/aten_reduce_probe.txt" | tail -12
