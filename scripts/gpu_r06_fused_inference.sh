#!/bin/bash
# Round 6, second part: the one-launch inference pass (cusrl_mlp2_forward) — its tests, the bench A/B against the library GEMM
# chain (CUSRL_FUSED_INFERENCE=0) interleaved on one box, the per-region device floor.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06b
mkdir -p "$O"
cd "$R"
export TMPDIR=/tmp
B="--no-cpu-baseline --no-kernel-pass --no-scale-pass --no-env-ab"
brief() { python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$1', d['value'], d['ms_per_step'], d['ppo_update_ms'])"; }
timeout 600 python -m pytest tests/test_mlp_forward.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15 | tee "$O/pytest_mlp_forward.txt"
for i in 1 2 3; do
  python bench.py $B 2>/dev/null | tail -1 | brief "default (fused inference pass)"
  CUSRL_FUSED_INFERENCE=0 python bench.py $B 2>/dev/null | tail -1 | brief "CUSRL_FUSED_INFERENCE=0"
done | tee "$O/fused_inference_ab.txt"
python scripts/graph_floor.py 2>&1 | grep -v amdgpu.ids | tee "$O/graph_floor_fused.txt"
