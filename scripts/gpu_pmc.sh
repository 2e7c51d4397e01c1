#!/bin/bash
# HBM traffic of the gather / pack launches from PMC counters, as MI355X_MICROARCH.md §HBM prescribes: FETCH_SIZE and
# WRITE_SIZE in SEPARATE rocprofv3 passes (TCC slot limit), --kernel-trace only.  Usage: scripts/gpu_pmc.sh [tag]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-pmc}
SAFE=${TAG//\//_}  # (a tag may name a sub-directory of gpurun_out/)
cd /tmp && export TMPDIR=/tmp
mkdir -p "$R/gpurun_out/$TAG"
# third pass (round 3): SQ counters of the same launches — where the waves' cycles go (VALU issue, LDS, waiting), LDS bank
# conflicts; 8 SQ slots per pass (MI355X_MICROARCH.md "rocprofv3 PMC slots")
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_VALU"
for C in FETCH_SIZE WRITE_SIZE SQ; do
  OUT=/tmp/pmc_${SAFE}_$C; rm -rf "$OUT"
  if [ "$C" = SQ ]; then LIST="$SQ"; else LIST="$C"; fi
  timeout 400 rocprofv3 --pmc $LIST --kernel-trace --output-format csv -d "$OUT" -o pmc -- \
      python "$R/scripts/pmc_cases.py" "$R/gpurun_out/$TAG" > /tmp/pmc_${SAFE}_$C.log 2>&1 < /dev/null
  echo "$C pass rc=$?"
  F=$(find "$OUT" -name "*counter_collection.csv" < /dev/null | head -1)
  if [ -n "$F" ]; then
    head -1 "$F" > "$R/gpurun_out/$TAG/${C}_counters.csv"
    grep -E "cusrl::(gather_kernel|push_kernel|pack_rows_kernel|ppo_loss_rowgroup_kernel)" "$F" >> "$R/gpurun_out/$TAG/${C}_counters.csv"
    wc -l "$R/gpurun_out/$TAG/${C}_counters.csv"
  else echo "no counter csv"; find "$OUT" -type f < /dev/null | head; tail -5 /tmp/pmc_${SAFE}_$C.log; fi
done
python "$R/scripts/pmc_summarize.py" "$R/gpurun_out/$TAG" "$R/gpurun_out/$TAG/pmc_summary.json" | tail -60
