#!/bin/bash
# HIP runtime switches that shape how a hipGraph's parallel branches reach the hardware queues, one at a time against the
# default, interleaved on one box (bench workload).   Usage (through gpurun): bash scripts/gpu_r06_runtime_flags.sh
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06b
mkdir -p "$O"
cd "$R"
export TMPDIR=/tmp
B="--no-cpu-baseline --no-kernel-pass --no-scale-pass --no-env-ab --steps 40 --warmup 10"
brief() { python -c "
import sys, json
ok = False
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$1', d['value'], d['ms_per_step'], d['ppo_update_ms']); ok = True
if not ok: print('$1', 'FAILED')"; }
for i in 1 2; do
  python bench.py $B 2>/dev/null | tail -1 | brief "default"
  for flag in DEBUG_HIP_FORCE_GRAPH_QUEUES=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=2 DEBUG_HIP_FORCE_GRAPH_QUEUES=8 GPU_MAX_HW_QUEUES=2 GPU_MAX_HW_QUEUES=8 \
              DEBUG_HIP_DYNAMIC_QUEUES=0 DEBUG_HIP_DYNAMIC_QUEUES=1 GPU_STREAMOPS_CP_WAIT=0 GPU_STREAMOPS_CP_WAIT=1 DEBUG_HIP_GRAPH_BATCH_SIZE=1024 AMD_DIRECT_DISPATCH=0; do
    env $flag timeout 200 python bench.py $B 2>/dev/null | tail -1 | brief "$flag"
  done
done | tee "$O/runtime_flags_ab.txt"
