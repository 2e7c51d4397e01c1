#!/bin/bash
# Round-4 PMC passes over scripts/pmc_r04_cases.py: one rocprofv3 run per counter set (--kernel-trace only, as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in separate passes).  Usage: scripts/gpu_pmc_r04.sh [tag]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04/pmc}
cd /tmp && export TMPDIR=/tmp
mkdir -p "$R/gpurun_out/$TAG"
declare -A SETS
SETS[fetch]="FETCH_SIZE"
SETS[write]="WRITE_SIZE"
SETS[l2]="TCC_HIT_sum TCC_MISS_sum"
SETS[ea_read]="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum"
SETS[ea_write]="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum"
SETS[stall]="TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum"
SETS[sq]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VMEM GRBM_GUI_ACTIVE"
SETS[sq2]="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU"
SETS[latency]="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum"
for NAME in fetch write l2 ea_read ea_write stall sq sq2 latency; do
  OUT=/tmp/pmc_r04_$NAME; rm -rf "$OUT"
  timeout 300 rocprofv3 --pmc ${SETS[$NAME]} --kernel-trace --output-format csv -d "$OUT" -o pmc -- \
      python "$R/scripts/pmc_r04_cases.py" "$R/gpurun_out/$TAG" > /tmp/pmc_r04_$NAME.log 2>&1 < /dev/null
  echo "$NAME pass rc=$?"
  F=$(find "$OUT" -name "*counter_collection.csv" < /dev/null | head -1)
  if [ -n "$F" ]; then
    head -1 "$F" > "$R/gpurun_out/$TAG/${NAME}_counters.csv"
    grep -E "cusrl::(gae_kernel|next_value_kernel|normalize_from_partials_kernel|push_kernel|narrow_linear_bwd_kernel|colsum_chunked_kernel)" "$F" >> "$R/gpurun_out/$TAG/${NAME}_counters.csv"
    wc -l "$R/gpurun_out/$TAG/${NAME}_counters.csv"
  else echo "no counter csv"; tail -5 /tmp/pmc_r04_$NAME.log; fi
done
python "$R/scripts/pmc_r04_summarize.py" "$R/gpurun_out/$TAG" "$R/gpurun_out/$TAG/pmc_summary.json"
