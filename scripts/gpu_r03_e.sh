#!/bin/bash
mkdir -p gpurun_out/r03
python -m pytest tests/test_hip_kernels.py tests/test_agent_gpu.py tests/test_baseline_configs.py -m gpu -q --timeout 600 -k "loss or trace or hipgraph or config1 or std_vector" 2>&1 | grep -v "tensor(\[" | tail -40 > gpurun_out/r03/test_e.log
tail -8 gpurun_out/r03/test_e.log
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/loss_bench_e.txt
import sys; sys.path.insert(0, "scripts")
import kernel_bench
for N in (4096, 1 << 20):
    for name, (us, nbytes) in kernel_bench.bench_size(N, only=("ppo loss",), iters=200 if N == 4096 else 20).items():
        print(f"N={N:8d} {name:50s} {us:9.2f} us  {nbytes/us/1e3:8.1f} GB/s  {nbytes/us/1e3/8000:.3f}")
PY
