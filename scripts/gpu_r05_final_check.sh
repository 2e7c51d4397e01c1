#!/bin/bash
# smoke() + the whole GPU suite as the driver runs it (-x), summary line kept; the distributed tests a second time
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05final2
mkdir -p "$O"
cd "$R"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > "$O/smoke.txt"
python -m pytest tests -x -q -m gpu > "$O/pytest_full.txt" 2>&1
grep -E "passed|failed|error" "$O/pytest_full.txt" | tail -3 > "$O/pytest_gpu.txt"
python -m pytest tests/test_distributed_gpu.py -q -m gpu > "$O/pytest_dist_again.txt" 2>&1
grep -E "passed|failed|error" "$O/pytest_dist_again.txt" | tail -2 >> "$O/pytest_gpu.txt"
cat "$O/smoke.txt" "$O/pytest_gpu.txt"; grep -B2 -A30 "^FAILED\|^E  " "$O/pytest_full.txt" "$O/pytest_dist_again.txt" | cut -c1-400 | head -80
