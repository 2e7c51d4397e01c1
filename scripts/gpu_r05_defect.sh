#!/bin/bash
# First GPU call of the next session (≈ 1 min): the probes DESIGN.md section 7 ("What comes next", item 4) lists for the single-stream defect.
#   gpurun --timeout 300 -- 'bash scripts/gpu_r05_defect.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_defect
mkdir -p "$OUT"
cd "$R"
export CUSRL_CONCURRENT_CRITIC=0
{
echo "== stock composition, single stream, in-graph snapshots"
DEBUG_KIND=continuous DEBUG_ITERATIONS=8 DEBUG_INGRAPH=1 python scripts/debug_amp_identity.py
echo "== AMP composition, single stream, in-graph snapshots"
DEBUG_INGRAPH=1 python scripts/debug_amp_identity.py
echo "== AMP composition, single stream, rocBLAS atomics off, no NaN fill"
DEBUG_DETERMINISTIC=2 python scripts/debug_amp_identity.py
echo "== AMP composition, single stream, untuned GEMM selection + rocBLAS atomics off"
CUSRL_TUNED_GEMMS=0 DEBUG_DETERMINISTIC=2 python scripts/debug_amp_identity.py
} 2>&1 | grep -v amdgpu.ids | cut -c1-300 > "$OUT/probes.txt"
tail -60 "$OUT/probes.txt"
