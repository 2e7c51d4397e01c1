#!/bin/bash
# kernel sequence of one captured env step of the bench workload (rocprofv3 kernel trace, scripts/kernel_sequence.py)
cd /tmp && export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-env_step}
mkdir -p $R/gpurun_out/$TAG
rm -rf /tmp/es; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/es -o bench -- python $R/bench.py --no-cpu-baseline --no-kernel-pass --no-scale-pass --no-env-ab --steps 6 --warmup 6 > /tmp/es.log 2>&1 < /dev/null
T=$(find /tmp/es -name "*kernel_trace.csv" | head -1)
python $R/scripts/kernel_sequence.py $T synthetic_env_step --nth -5 | tee $R/gpurun_out/$TAG/env_step_sequence.txt
