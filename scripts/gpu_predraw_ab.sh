#!/bin/bash
# Third part of round 6: the exploration noise of a whole rollout drawn ahead of it (CUSRL_PREDRAW_NOISE, default on) vs inside
# every captured env step, interleaved on one box; single process and one RCCL rank.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r06_predraw; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-kernel-pass --no-scale-pass --no-env-ab --steps 50 --warmup 10"
brief() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/it, update', d['ppo_update_ms'], 'ms,', round(d['value']/1e6,2), 'M env-steps/s')"; }
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
for i in 1 2 3; do
  python $R/bench.py $B 2>/dev/null | brief "single process, noise drawn ahead of the rollout :" | tee -a $OUT/ab.txt
  CUSRL_PREDRAW_NOISE=0 python $R/bench.py $B 2>/dev/null | brief "single process, noise drawn inside every step   :" | tee -a $OUT/ab.txt
done
$TR --master-port 29911 $R/bench.py --gpus 1 $B 2>/dev/null | tail -1 | brief "one rank, noise drawn ahead of the rollout       :" | tee -a $OUT/ab.txt
CUSRL_PREDRAW_NOISE=0 $TR --master-port 29912 $R/bench.py --gpus 1 $B 2>/dev/null | tail -1 | brief "one rank, noise drawn inside every step         :" | tee -a $OUT/ab.txt
