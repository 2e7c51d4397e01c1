#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05final2
mkdir -p "$O"
cd "$R"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > "$O/smoke.txt"
python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > "$O/pytest_gpu.txt"
bash scripts/gpu_pmc_r05.sh r05final2/pmc > "$O/pmc_log.txt" 2>&1
cat "$O/smoke.txt" "$O/pytest_gpu.txt"; tail -12 "$O/pmc_log.txt" | cut -c1-600
