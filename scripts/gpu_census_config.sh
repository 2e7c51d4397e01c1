#!/bin/bash
# Runs on the GPU box (via gpurun): per-iteration kernel census of one BASELINE config under hipGraph replay — two
# rocprofv3 kernel traces with different iteration counts, differenced so that warm-up and capture drop out.
# Usage: scripts/gpu_census_config.sh <config> [tag]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
CFG=${1:-config5}; TAG=${2:-census_$CFG}
cd /tmp && export TMPDIR=/tmp
mkdir -p "$R/gpurun_out/$TAG"
for N in 8 16; do
  rm -rf /tmp/census_$N
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/census_$N -o run -- \
      python "$R/scripts/run_config.py" "$CFG" --compile --iterations $N > /tmp/census_$N.log 2>&1 < /dev/null
  echo "rocprofv3 N=$N rc=$?"; grep iteration /tmp/census_$N.log | tail -1 | cut -c1-100
done
python3 - "$R/gpurun_out/$TAG/census.csv" <<'PY'
import csv, sys, collections, glob
def load(n):
    path = glob.glob(f"/tmp/census_{n}/**/*kernel_trace.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: [0, 0])
    with open(path) as f:
        for row in csv.DictReader(f):
            a = acc[row["Kernel_Name"].split("(")[0][:110]]
            a[0] += 1; a[1] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    return acc
a, b = load(8), load(16)
rows = []
for name, (n, t) in b.items():
    n0, t0 = a.get(name, (0, 0))
    if n > n0:
        rows.append((name, (n - n0) / 8, (t - t0) / 8 / 1e3))
rows.sort(key=lambda r: -r[2])
with open(sys.argv[1], "w") as f:
    f.write("kernel,calls_per_iteration,us_per_iteration\n")
    for r in rows:
        f.write(f"\"{r[0]}\",{r[1]:.1f},{r[2]:.1f}\n")
print(f"kernels per iteration {sum(r[1] for r in rows):.0f}, busy {sum(r[2] for r in rows) / 1e3:.2f} ms")
for r in rows[:45]:
    print(f"{r[1]:8.1f} {r[2]:9.1f} us  {r[0][:100]}")
PY
