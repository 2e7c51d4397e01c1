#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats of a short bench run; copies only the small
# CSV summaries into gpurun_out/<tag>/ (the raw trace stays in /tmp).  Usage: scripts/gpu_profile.sh <tag> [bench args]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-prof}; shift || true
OUT=/tmp/prof_$TAG
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT"
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o bench -- \
    python "$R/bench.py" --no-cpu-baseline --no-kernel-pass "$@" > /tmp/prof_$TAG.log 2>&1 < /dev/null
echo "rocprofv3 rc=$?"
grep "^{" /tmp/prof_$TAG.log | cut -c1-400
mkdir -p "$R/gpurun_out/$TAG"
find "$OUT" -name "*stats*.csv" -size -2000k -exec cp {} "$R/gpurun_out/$TAG/" \; < /dev/null
grep "^{" /tmp/prof_$TAG.log > "$R/gpurun_out/$TAG/bench_line.json" < /dev/null
ls -la "$R/gpurun_out/$TAG" < /dev/null
F=$(find "$OUT" -name "*kernel_stats.csv" < /dev/null | head -1)
if [ -n "$F" ]; then head -50 "$F" | cut -c1-220; else echo "no kernel_stats.csv"; find "$OUT" -type f < /dev/null | head; fi
