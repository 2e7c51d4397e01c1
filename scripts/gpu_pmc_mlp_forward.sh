#!/bin/bash
# HBM traffic of cusrl_mlp2_forward from PMC counters: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes (TCC slot limit),
# a third pass with SQ counters; --kernel-trace only (MI355X_MICROARCH.md, HBM / rocprofv3 section).  Usage: gpu_pmc_r06.sh [tag]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06_pmc_mlp}
SAFE=${TAG//\//_}
cd /tmp && export TMPDIR=/tmp
mkdir -p "$R/gpurun_out/$TAG"
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
for C in FETCH_SIZE WRITE_SIZE SQ; do
  OUT=/tmp/pmc_${SAFE}_$C; rm -rf "$OUT"
  if [ "$C" = SQ ]; then LIST="$SQ"; else LIST="$C"; fi
  timeout 400 rocprofv3 --pmc $LIST --kernel-trace --output-format csv -d "$OUT" -o pmc -- \
      python "$R/scripts/pmc_mlp_forward_cases.py" "$R/gpurun_out/$TAG" > /tmp/pmc_${SAFE}_$C.log 2>&1 < /dev/null
  echo "$C pass rc=$?"
  F=$(find "$OUT" -name "*counter_collection.csv" < /dev/null | head -1)
  if [ -n "$F" ]; then
    head -1 "$F" > "$R/gpurun_out/$TAG/${C}_counters.csv"
    grep -E "cusrl::(push_kernel|mlp2_forward_kernel)" "$F" >> "$R/gpurun_out/$TAG/${C}_counters.csv"
    wc -l "$R/gpurun_out/$TAG/${C}_counters.csv"
  else echo "no counter csv"; find "$OUT" -type f < /dev/null | head; tail -5 /tmp/pmc_${SAFE}_$C.log; fi
done
python "$R/scripts/pmc_mlp_forward_summarize.py" "$R/gpurun_out/$TAG" "$R/gpurun_out/$TAG/pmc_summary.json" | tail -60
