#!/bin/bash
# Round 6: what one RCCL rank costs over the single-process loop (the fixed overhead under any weak-scaling efficiency).
# Same box: the single-process line, the one-rank torchrun line (C-ABI route), and the one-rank run's idle-gap census.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r06_one_rank; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-kernel-pass --no-scale-pass --no-env-ab --steps 50 --warmup 10"
brief() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/it, update', d['ppo_update_ms'], 'ms,', round(d['value']/1e6,2), 'M env-steps/s;', d['config'].get('collectives'), d['config'].get('gradient_allreduce'))"; }
for i in 1 2 3; do
  python $R/bench.py $B 2>/dev/null | brief "single process :" | tee -a $OUT/bench_one_rank.txt
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$i $R/bench.py --gpus 1 $B 2>/dev/null | tail -1 | brief "one RCCL rank  :" | tee -a $OUT/bench_one_rank.txt
done
rm -rf /tmp/one_rank_trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/one_rank_trace -o bench -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 $R/bench.py --gpus 1 --no-cpu-baseline --no-kernel-pass --no-scale-pass --no-env-ab --steps 6 --warmup 6 > /tmp/one_rank.log 2>&1 < /dev/null
for T in $(find /tmp/one_rank_trace -name "*kernel_trace.csv"); do
  n=$(wc -l < $T); echo "$T $n"
  if [ $n -gt 1000 ]; then
    python $R/scripts/idle_gaps.py $T --min-us 10 > $OUT/idle_gaps_one_rank.txt
    python $R/scripts/step_timeline.py $T "ppo_loss_rowgroup" --nth -7 --steps 2 > $OUT/timeline_one_rank.txt
  fi
done
