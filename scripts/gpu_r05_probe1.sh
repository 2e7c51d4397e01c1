#!/bin/bash
# Round 5, GPU probe of the captured-step defect: (a) plain-torch probe of captured ATen global reduces, (b) the single-stream
# identity probe with a node census of its captured steps, as shipped in round 4 (CUSRL_WIDE_LINEAR_MIN_ROWS=4096) and with
# every linear backward on the repo's HIP column sums, (c) node census of the BASELINE configs.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_probe1
mkdir -p "$OUT"
cd "$R"
timeout 600 python scripts/probe_aten_reduce_capture.py ${PROBE_REPLAYS:-2000} 2>&1 | grep -v amdgpu.ids > "$OUT/aten_reduce_probe.txt"
export CUSRL_CONCURRENT_CRITIC=0
for kind in amp continuous; do
  for rows in 4096 1; do
    echo "== $kind single stream, CUSRL_WIDE_LINEAR_MIN_ROWS=$rows"
    if [ $kind = continuous ]; then export DEBUG_KIND=continuous DEBUG_ITERATIONS=8; else unset DEBUG_KIND; export DEBUG_ITERATIONS=4; fi
    # (rows=4096 is round 4's configuration: it only reproduces the defect with the memset nodes left in place)
    CUSRL_GRAPH_MEMSETS=keep CUSRL_WIDE_LINEAR_MIN_ROWS=$rows timeout 300 python scripts/debug_amp_identity.py 2>&1 | tail -40
  done
done 2>&1 | grep -v amdgpu.ids | cut -c1-400 > "$OUT/identity.txt"
unset CUSRL_CONCURRENT_CRITIC DEBUG_KIND DEBUG_ITERATIONS
for c in config1 config2 config5; do
  echo "== census $c (shipped default)"; timeout 300 python scripts/graph_census.py $c 2>&1 | tail -45
done 2>&1 | grep -v amdgpu.ids | cut -c1-300 > "$OUT/census.txt"
tail -30 "$OUT/aten_reduce_probe.txt"; tail -60 "$OUT/identity.txt"; cat "$OUT/census.txt"
