#!/bin/bash
# round 3, call F: full GPU suite after the record rework, gather / pack / push timings at both sizes, bench
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -q --timeout 600 2>&1 | grep -v "tensor(\[" | tail -80 > gpurun_out/r03/test_f.log
tail -12 gpurun_out/r03/test_f.log
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/record_bench_f.txt
import sys; sys.path.insert(0, "scripts")
import kernel_bench
for N in (4096, 1 << 20):
    for name, (us, nbytes) in kernel_bench.bench_size(N, only=("gather", "pack", "push"), iters=200 if N == 4096 else 10).items():
        print(f"N={N:8d} {name:95s} {us:9.2f} us  {nbytes/us/1e3:8.1f} GB/s  {nbytes/us/1e3/8000:.3f}")
PY
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-kernel-pass --no-scale-pass 2>/dev/null | tail -1 > gpurun_out/r03/bench_f_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03/bench_f_?.json')):
    d=json.loads(open(f).read()); print(f, d['value'], d['ms_per_step'], d['ppo_update_ms'], d['roofline']['avg_us'], d['roofline']['frac'])
PY
