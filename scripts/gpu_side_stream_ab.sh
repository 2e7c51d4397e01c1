#!/bin/bash
# Second part of round 6: the draw-ahead stream as a probed HIGH-priority stream (CUSRL_SIDE_STREAM_PRIORITY=1) vs the probed normal-priority one (default), single process and
# one RCCL rank, interleaved on one box (run-to-run spread included: the queue a normal-priority stream lands on can also be the one
# a replayed graph's executor uses for its second branch, which no probe can see).
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r06_side_stream; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-kernel-pass --no-scale-pass --no-env-ab --steps 50 --warmup 10"
brief() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/it, update', d['ppo_update_ms'], 'ms,', round(d['value']/1e6,2), 'M env-steps/s')"; }
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
for i in 1 2 3 4; do
  CUSRL_SIDE_STREAM_PRIORITY=1 python $R/bench.py $B 2>/dev/null | brief "single process, high-priority draw stream   :" | tee -a $OUT/ab.txt
  python $R/bench.py $B 2>/dev/null | brief "single process, normal-priority draw stream :" | tee -a $OUT/ab.txt
  CUSRL_SIDE_STREAM_PRIORITY=1 $TR --master-port 2971$i $R/bench.py --gpus 1 $B 2>/dev/null | tail -1 | brief "one rank, high-priority draw stream         :" | tee -a $OUT/ab.txt
  $TR --master-port 2972$i $R/bench.py --gpus 1 $B 2>/dev/null | tail -1 | brief "one rank, normal-priority draw stream       :" | tee -a $OUT/ab.txt
done
