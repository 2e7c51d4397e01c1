#!/bin/bash
# smoke() + the whole GPU suite as the driver runs it (-x), summary line kept; configs 1 / 2 / 5 and config 5's node census
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05final4
mkdir -p "$O"
cd "$R"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > "$O/smoke.txt"
python -m pytest tests -x -q -m gpu > "$O/pytest_full.txt" 2>&1
grep -E "passed|failed|error" "$O/pytest_full.txt" | tail -3 > "$O/pytest_gpu.txt"
for c in config1 config5; do for i in 1 2; do echo "$c $(timeout 300 python scripts/run_config.py $c --compile --iterations 8 2>&1 | grep '^iteration' | tail -1 | cut -c1-60)"; done; done > "$O/configs.txt"
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-kernel-pass --no-scale-pass --no-env-ab 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config2', d['ms_per_step'], d['ppo_update_ms'])"; done >> "$O/configs.txt"
timeout 300 python scripts/graph_census.py config5 2>&1 | grep -v amdgpu.ids | tail -14 | cut -c1-220 > "$O/census5.txt"
cat "$O/smoke.txt" "$O/pytest_gpu.txt" "$O/configs.txt" "$O/census5.txt"; grep -h -A14 "^FAILED\|AssertionError\|Error" "$O/pytest_full.txt" | cut -c1-400 | head -50
