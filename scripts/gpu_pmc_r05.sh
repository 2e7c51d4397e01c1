#!/bin/bash
# Round-5 PMC passes over scripts/pmc_r05_cases.py (the PPO objective, both scalar-stream layouts): one rocprofv3 run per
# counter set (--kernel-trace only, FETCH_SIZE and WRITE_SIZE in separate passes as MI355X_MICROARCH.md prescribes).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05/pmc}
cd /tmp && export TMPDIR=/tmp
mkdir -p "$R/gpurun_out/$TAG"
declare -A SETS
SETS[fetch]="FETCH_SIZE"
SETS[write]="WRITE_SIZE"
SETS[sq]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VMEM GRBM_GUI_ACTIVE"
SETS[sq2]="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU"
SETS[latency]="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum"
SETS[l2]="TCC_HIT_sum TCC_MISS_sum"
SETS[ea_read]="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum"
SETS[ea_write]="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum"
for NAME in fetch write sq sq2 latency l2 ea_read ea_write; do
  OUT=/tmp/pmc_r05_$NAME; rm -rf "$OUT"
  timeout 300 rocprofv3 --pmc ${SETS[$NAME]} --kernel-trace --output-format csv -d "$OUT" -o pmc -- \
      python "$R/scripts/pmc_r05_cases.py" "$R/gpurun_out/$TAG" > /tmp/pmc_r05_$NAME.log 2>&1 < /dev/null
  echo "$NAME pass rc=$?"
  F=$(find "$OUT" -name "*counter_collection.csv" < /dev/null | head -1)
  if [ -n "$F" ]; then
    head -1 "$F" > "$R/gpurun_out/$TAG/${NAME}_counters.csv"
    grep -E "cusrl::ppo_loss_rowgroup_kernel" "$F" >> "$R/gpurun_out/$TAG/${NAME}_counters.csv"
    wc -l "$R/gpurun_out/$TAG/${NAME}_counters.csv"
  else echo "no counter csv"; tail -5 /tmp/pmc_r05_$NAME.log; fi
done
python "$R/scripts/pmc_r04_summarize.py" "$R/gpurun_out/$TAG" "$R/gpurun_out/$TAG/pmc_summary.json"
