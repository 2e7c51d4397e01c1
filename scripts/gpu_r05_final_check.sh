#!/bin/bash
# smoke() + the whole GPU suite as the driver runs it (-x), summary line kept; the distributed tests two more times
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05final2
mkdir -p "$O"
cd "$R"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > "$O/smoke.txt"
python -m pytest tests -x -q -m gpu > "$O/pytest_full.txt" 2>&1
grep -E "passed|failed|error" "$O/pytest_full.txt" | tail -3 > "$O/pytest_gpu.txt"
for i in 1 2; do python -m pytest tests/test_distributed_gpu.py -q -m gpu > "$O/pytest_dist_again$i.txt" 2>&1; grep -E "passed|failed" "$O/pytest_dist_again$i.txt" | tail -1 >> "$O/pytest_gpu.txt"; done
cat "$O/smoke.txt" "$O/pytest_gpu.txt"; grep -h -A12 "^FAILED\|AssertionError\|what():" "$O/pytest_full.txt" "$O"/pytest_dist_again*.txt | cut -c1-500 | head -60
