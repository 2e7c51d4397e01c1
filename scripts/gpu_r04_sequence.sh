#!/bin/bash
# kernel sequences of one captured env step and one captured minibatch step of the bench workload
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r04/sequence}; shift || true
OUT=/tmp/seq; rm -rf $OUT; cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$TAG
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o bench -- python $R/bench.py --no-cpu-baseline --no-kernel-pass --no-scale-pass --steps 6 --warmup 6 "$@" > /tmp/seq.log 2>&1 < /dev/null
grep "^{" /tmp/seq.log | cut -c1-300
T=$(find $OUT -name "*kernel_trace.csv" | head -1)
python $R/scripts/kernel_sequence.py $T normal_sample_logp --nth -5 | tee $R/gpurun_out/$TAG/env_step_sequence.txt
python $R/scripts/kernel_sequence.py $T "ppo_loss_rowgroup" --nth -3 | tee $R/gpurun_out/$TAG/minibatch_step_sequence.txt
