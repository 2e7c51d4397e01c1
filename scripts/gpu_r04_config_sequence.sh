#!/bin/bash
# kernel sequence of one env step / one minibatch step of a BASELINE config (scripts/run_config.py) + per-iteration census
# usage: scripts/gpu_r04_config_sequence.sh config5 [tag] [step anchor] [update anchor]
R=${GRAFT_REPO_ROOT:-/root/repo}; CFG=${1:-config5}; TAG=${2:-r04/$CFG}; A1=${3:-normal_sample_logp}; A2=${4:-ppo_loss_rowgroup}
OUT=/tmp/seq_$CFG; rm -rf $OUT; cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$TAG
COMPILE=--compile; [ "$CFG" = config4 ] && COMPILE=
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- python $R/scripts/run_config.py $CFG $COMPILE --iterations 8 > $R/gpurun_out/$TAG/run.txt 2>&1 < /dev/null
grep iteration $R/gpurun_out/$TAG/run.txt | tail -3
T=$(find $OUT -name "*kernel_trace.csv" | head -1)
python $R/scripts/kernel_sequence.py $T "$A1" --nth -5 > $R/gpurun_out/$TAG/env_step_sequence.txt
python $R/scripts/kernel_sequence.py $T "$A2" --nth -3 > $R/gpurun_out/$TAG/minibatch_step_sequence.txt
python3 - $T $R/gpurun_out/$TAG/census.txt <<'PY'
import csv, sys, re, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# last complete iteration = between the last two launches of the GAE kernel
marks = [i for i, r in enumerate(rows) if "gae_kernel" in r["Kernel_Name"]]
a, b = marks[-2], marks[-1]
acc = collections.defaultdict(lambda: [0, 0])
for r in rows[a:b]:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:100]
    acc[name][0] += 1; acc[name][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
total = sum(v[0] for v in acc.values()); busy = sum(v[1] for v in acc.values())
aten = sum(v[0] for k, v in acc.items() if k.startswith("at::") or "rocclr" in k or k.startswith("rocprim") or "elementwise" in k)
with open(sys.argv[2], "w") as f:
    f.write(f"one iteration (between the last two GAE launches): {total} kernels, {busy/1e6:.2f} ms busy, wall {(int(rows[b]['Start_Timestamp'])-int(rows[a]['Start_Timestamp']))/1e6:.2f} ms; aten / rocprim / copy launches: {aten}\n")
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        f.write(f"{v[0]:6d} {v[1]/1e3:9.1f} us  {k}\n")
print(open(sys.argv[2]).readline())
PY
