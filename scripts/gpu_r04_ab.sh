#!/bin/bash
# interleaved A/B of the bench line on ONE box: scripts/gpu_r04_ab.sh VAR "a b" [rounds] [extra bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}; VAR=$1; VALUES=$2; ROUNDS=${3:-3}; shift 3 || true
for i in $(seq $ROUNDS); do for V in $VALUES; do
  env $VAR=$V python $R/bench.py --no-cpu-baseline --no-kernel-pass --no-scale-pass "$@" 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$V', d['ms_per_step'], 'ms/iteration  update', d['ppo_update_ms'])"
done; done
