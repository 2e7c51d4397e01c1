#!/bin/bash
# Second part of round 6: what runs between the start of pre_update and the first minibatch step, and between the last optimizer
# step and the next rollout, single process vs one RCCL rank (rocprofv3 kernel traces of bench.py --steps 6 --warmup 6).
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r06_one_rank_sequence; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-kernel-pass --no-scale-pass --no-env-ab --steps 6 --warmup 6"
rm -rf /tmp/seq_single /tmp/seq_rank
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/seq_single -o bench -- python $R/bench.py $B > /tmp/seq_single.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/seq_rank -o bench -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29651 $R/bench.py --gpus 1 $B > /tmp/seq_rank.log 2>&1 < /dev/null
for tag in single rank; do
  for T in $(find /tmp/seq_$tag -name "*kernel_trace.csv"); do
    n=$(wc -l < $T)
    if [ $n -gt 1000 ]; then
      python $R/scripts/window_timeline.py $T count_flags_kernel ppo_loss_rowgroup --nth -2 --before 6 > $OUT/pre_update_$tag.txt
      python $R/scripts/window_timeline.py $T policy_stats count_flags_kernel --nth -2 --before 12 | head -60 > $OUT/after_update_$tag.txt
    fi
  done
done
wc -l $OUT/*
