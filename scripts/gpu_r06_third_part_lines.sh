#!/bin/bash
# Third part of round 6, final tree: the bench line (default flags, the driver's flags), repeats, the one-rank pairs + trace and the
# rocprofv3 kernel stats of the same command — one box.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06third; mkdir -p $O
cd $R; export TMPDIR=/tmp
B="--no-cpu-baseline --no-kernel-pass --no-scale-pass --no-env-ab"
brief() { python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$1', d['value'], d['ms_per_step'], d['ppo_update_ms'])"; }
python bench.py 2>/dev/null | tail -1 > $O/bench_line.json
python -c "import json;d=json.load(open('$O/bench_line.json'));print('bench', d['value'], d['ms_per_step'], d['ppo_update_ms'], d['roofline']['frac'], d['cpu_baseline']['value'])"
for i in 1 2 3; do python bench.py $B 2>/dev/null | tail -1; done | brief repeat | tee $O/bench_repeats.txt
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_line_driver_flags.json
python -c "import json;d=json.load(open('$O/bench_line_driver_flags.json'));print('driver flags', d['value'], d['ms_per_step'], d['ppo_update_ms'])" | tee -a $O/bench_repeats.txt
rm -rf gpurun_out/r06_one_rank; bash scripts/gpu_r06_one_rank.sh > $O/one_rank.log 2>&1; cp gpurun_out/r06_one_rank/* $O/; cat $O/bench_one_rank.txt | cut -c1-90
bash scripts/gpu_profile.sh r06third/prof --steps 20 --warmup 6 > $O/gpu_profile.log 2>&1; tail -4 $O/gpu_profile.log | cut -c1-200
