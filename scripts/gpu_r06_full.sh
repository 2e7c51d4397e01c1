#!/bin/bash
# Round 6: the whole GPU suite, then the bench line with the driver's flags.  Output under gpurun_out/r06_full/.
set -u
OUT=gpurun_out/r06_full
mkdir -p $OUT
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=15 > $OUT/pytest_gpu.txt 2>&1
tail -25 $OUT/pytest_gpu.txt
python bench.py --no-cpu-baseline --no-kernel-pass --no-scale-pass --no-env-ab --steps 50 --warmup 10 2>$OUT/bench_err.txt | tee $OUT/bench_short.json | cut -c1-300
