#!/bin/bash
mkdir -p gpurun_out/r03/configs
python -m pytest tests -m gpu -q --timeout 600 2>&1 | grep -v "tensor(\[" | tail -60 > gpurun_out/r03/test_h.log
tail -8 gpurun_out/r03/test_h.log
timeout 300 python scripts/run_config.py config4 --iterations 5 2>&1 | grep -v amdgpu.ids > gpurun_out/r03/configs/run_config4.txt; grep iteration gpurun_out/r03/configs/run_config4.txt | tail -2 | cut -c1-150
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_c4 && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c4 -o c4 -- python $GRAFT_REPO_ROOT/scripts/run_config.py config4 --iterations 4 > /tmp/c4.log 2>&1
F=$(find /tmp/prof_c4 -name "*kernel_stats.csv" | head -1); cp $F $GRAFT_REPO_ROOT/gpurun_out/r03/config4_kernel_stats.csv; head -25 $F | cut -c1-160
