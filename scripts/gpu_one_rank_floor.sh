#!/bin/bash
# Second part of round 6: what one RCCL rank costs over the single-process loop, switch by switch, interleaved on one box; then
# the captured regions' floors and the boundaries between them (scripts/graph_floor.py) in both kinds of process.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r06_one_rank_floor; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-kernel-pass --no-scale-pass --no-env-ab --steps 50 --warmup 10"
brief() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/it, update', d['ppo_update_ms'], 'ms,', round(d['value']/1e6,2), 'M env-steps/s')"; }
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
for i in 1 2; do
  python $R/bench.py $B 2>/dev/null | brief "single process                                  :" | tee -a $OUT/ab.txt
  $TR --master-port 2961$i $R/bench.py --gpus 1 $B 2>/dev/null | tail -1 | brief "one rank (default)                              :" | tee -a $OUT/ab.txt
  CUSRL_SIDE_STREAM_PROBE=0 $TR --master-port 2962$i $R/bench.py --gpus 1 $B 2>/dev/null | tail -1 | brief "one rank, draw-ahead stream taken on faith       :" | tee -a $OUT/ab.txt
  CUSRL_NORMED_MAIN_FIRST=0 $TR --master-port 2963$i $R/bench.py --gpus 1 $B 2>/dev/null | tail -1 | brief "one rank, critic step launch captured first      :" | tee -a $OUT/ab.txt
  CUSRL_TWO_WINDOW_STEP=0 $TR --master-port 2964$i $R/bench.py --gpus 1 $B 2>/dev/null | tail -1 | brief "one rank, joined step                            :" | tee -a $OUT/ab.txt
done
echo "== single process" > $OUT/floor.txt
python $R/scripts/graph_floor.py >> $OUT/floor.txt 2>&1
echo "== one RCCL rank" >> $OUT/floor.txt
$TR --master-port 29631 $R/scripts/graph_floor.py >> $OUT/floor.txt 2>&1
grep -v "^\[W\|Warning\|warn\|amdgpu.ids\|version\|Hostname\|Librccl\|Gloo" $OUT/floor.txt | tail -50
