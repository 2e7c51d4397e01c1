#!/bin/bash
# Round 6, final tree: the pieces of profiles/r06/ that read a rocprofv3 kernel trace through an anchor kernel (the act step is
# cusrl_mlp2_forward since the second part: scripts/idle_gaps.py anchors on the env's step kernel), then the bench line itself
# (it quotes roofline.frac from the committed per-grid CSV).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06final; mkdir -p $O/timeline
bash $R/scripts/gpu_r06_timeline.sh r06final/timeline "update 1 tail" > $O/timeline.log 2>&1
bash $R/scripts/gpu_env_step_sequence.sh r06final/timeline > $O/env_step.log 2>&1
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/one_rank_trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/one_rank_trace -o bench -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 $R/bench.py --gpus 1 --no-cpu-baseline --no-kernel-pass --no-scale-pass --no-env-ab --steps 6 --warmup 6 > /tmp/one_rank.log 2>&1 < /dev/null
for T in $(find /tmp/one_rank_trace -name "*kernel_trace.csv"); do
  n=$(wc -l < $T)
  if [ $n -gt 1000 ]; then python $R/scripts/idle_gaps.py $T --min-us 10 > $O/idle_gaps_one_rank.txt; fi
done
cd $R
python bench.py 2>/dev/null | tail -1 > $O/bench_line.json
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_line_driver_flags.json
python -c "
import json
for f in ('bench_line.json', 'bench_line_driver_flags.json'):
    d = json.load(open('$O/' + f)); print(f, d['value'], d['ms_per_step'], d['ppo_update_ms'], d['roofline']['frac'], d['roofline']['avg_us'], d['cpu_baseline']['value'])"
head -12 $O/timeline/idle_gaps_epochsupdate_value1_prefetchtail.txt; head -4 $O/idle_gaps_one_rank.txt; cat $O/timeline/env_step_sequence.txt
