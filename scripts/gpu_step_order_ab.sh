#!/bin/bash
# Third part of round 6: which window's step launch an unjoined step captures first, single process (interleaved on one box).
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r06_step_order; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-kernel-pass --no-scale-pass --no-env-ab --steps 50 --warmup 10"
brief() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/it, update', d['ppo_update_ms'], 'ms,', round(d['value']/1e6,2), 'M env-steps/s')"; }
for i in 1 2 3 4; do
  python $R/bench.py $B 2>/dev/null | brief "single process, critic window captured first :" | tee -a $OUT/ab.txt
  CUSRL_STEP_MAIN_FIRST=1 python $R/bench.py $B 2>/dev/null | brief "single process, main window captured first   :" | tee -a $OUT/ab.txt
done
