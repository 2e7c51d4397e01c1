#!/bin/bash
# Round 5, second GPU call: probe with the memset-node replacement, the GPU suite on the new tree, stream-branch A/B.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_call2
mkdir -p "$OUT"
cd "$R"
PROBE_QUICK=1 timeout 400 python scripts/probe_aten_reduce_capture.py 1000 2>&1 | grep -v amdgpu.ids > "$OUT/aten_reduce_probe_quick.txt"
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -40 > "$OUT/pytest_gpu.txt"
for cc in unset 0 1; do
  if [ $cc = unset ]; then unset CUSRL_CONCURRENT_CRITIC; else export CUSRL_CONCURRENT_CRITIC=$cc; fi
  echo "== CUSRL_CONCURRENT_CRITIC=$cc bench"; timeout 300 python bench.py 2>&1 | tail -1 | cut -c1-700
  for c in config1 config5; do echo "== CUSRL_CONCURRENT_CRITIC=$cc $c"; timeout 300 python scripts/run_config.py $c --compile 2>&1 | tail -3; done
done 2>&1 | grep -v amdgpu.ids > "$OUT/stream_ab.txt"
cat "$OUT/aten_reduce_probe_quick.txt" | cut -c1-600; cat "$OUT/pytest_gpu.txt"; cat "$OUT/stream_ab.txt"
