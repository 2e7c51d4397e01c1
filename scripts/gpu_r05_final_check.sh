#!/bin/bash
# smoke() + the whole GPU suite as the driver runs it (-x), summary line kept
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05final5
mkdir -p "$O"
cd "$R"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > "$O/smoke.txt"
python -m pytest tests -x -q -m gpu > "$O/pytest_full.txt" 2>&1
grep -E "passed|failed|error" "$O/pytest_full.txt" | tail -3 > "$O/pytest_gpu.txt"
cat "$O/smoke.txt" "$O/pytest_gpu.txt"; grep -h -A14 "^FAILED\|AssertionError\|Error" "$O/pytest_full.txt" | cut -c1-400 | head -50
