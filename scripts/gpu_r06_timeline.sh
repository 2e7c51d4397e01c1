#!/bin/bash
# timeline (all queues) of consecutive minibatch steps of the bench workload, per setting of the round-6 switches
# usage: gpu_r06_timeline.sh <tag> "<EPOCH_GRAPHS> <SEPARATE_VALUE> <PREFETCH_GATHER>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r06_timeline}; shift
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$TAG
for cfg in "$@"; do
  set -- $cfg
  OUT=/tmp/seq_$1$2$3; rm -rf $OUT
  CUSRL_EPOCH_GRAPHS=$1 CUSRL_SEPARATE_VALUE_TERM=$2 CUSRL_PREFETCH_GATHER=$3 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o bench -- python $R/bench.py --no-cpu-baseline --no-kernel-pass --no-scale-pass --no-env-ab --steps 6 --warmup 6 > /tmp/seq.log 2>&1 < /dev/null
  grep "^{" /tmp/seq.log | cut -c1-200
  T=$(find $OUT -name "*kernel_trace.csv" | head -1)
  python $R/scripts/step_timeline.py $T "ppo_loss_rowgroup" --nth -7 --steps 3 --periods 45 > $R/gpurun_out/$TAG/timeline_epochs$1_value$2_prefetch$3.txt
  python $R/scripts/idle_gaps.py $T --min-us 10 > $R/gpurun_out/$TAG/idle_gaps_epochs$1_value$2_prefetch$3.txt
done
