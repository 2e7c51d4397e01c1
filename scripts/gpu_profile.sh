#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats of a short bench run; copies only the small
# CSV summaries into gpurun_out/<tag>/ (the raw trace stays in /tmp).  Usage: scripts/gpu_profile.sh <tag> [bench args]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-prof}; shift || true
SAFE=${TAG//\//_}  # (a tag may name a sub-directory of gpurun_out/)
OUT=/tmp/prof_$SAFE
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT"
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o bench -- \
    python "$R/bench.py" --no-cpu-baseline --no-kernel-pass --no-scale-pass "$@" > /tmp/prof_$SAFE.log 2>&1 < /dev/null
echo "rocprofv3 rc=$?"
grep "^{" /tmp/prof_$SAFE.log | cut -c1-400
mkdir -p "$R/gpurun_out/$TAG"
find "$OUT" -name "*stats*.csv" -size -2000k -exec cp {} "$R/gpurun_out/$TAG/" \; < /dev/null
grep "^{" /tmp/prof_$SAFE.log > "$R/gpurun_out/$TAG/bench_line.json" < /dev/null
ls -la "$R/gpurun_out/$TAG" < /dev/null
F=$(find "$OUT" -name "*kernel_stats.csv" < /dev/null | head -1)
if [ -n "$F" ]; then head -50 "$F" | cut -c1-220; else echo "no kernel_stats.csv"; find "$OUT" -type f < /dev/null | head; fi
# per (kernel, grid size) averages of the cusrl kernels from the raw dispatch trace (the stats CSV averages mix sizes)
T=$(find "$OUT" -name "*kernel_trace.csv" < /dev/null | head -1)
if [ -n "$T" ]; then
python3 - "$T" "$R/gpurun_out/$TAG/cusrl_kernels_by_grid.csv" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0, 10**18, 0])
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        name = row["Kernel_Name"]
        if "cusrl::" not in name:
            continue
        key = (name.split("(")[0], int(row.get("Grid_Size", row.get("Grid_Size_X", 0))))
        d = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
        a = acc[key]; a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
with open(sys.argv[2], "w") as f:
    f.write("kernel,grid_size_threads,calls,avg_ns,min_ns,max_ns\n")
    for (name, grid), (n, tot, lo, hi) in sorted(acc.items()):
        f.write(f"\"{name}\",{grid},{n},{tot / n:.0f},{lo},{hi}\n")
print(open(sys.argv[2]).read())
# stream occupancy: how much of the traced span the GPU was executing kernels, and where the idle time sits
spans = []
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        spans.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
spans.sort()
busy, gaps, cursor = 0, collections.Counter(), spans[0][0]
for start, end in spans:
    if start > cursor:
        gap = start - cursor
        bucket = "<5us" if gap < 5e3 else "5-20us" if gap < 2e4 else "20-100us" if gap < 1e5 else ">100us"
        gaps[bucket] += gap
        cursor = start
    if end > cursor:
        busy += end - cursor
        cursor = end
total = spans[-1][1] - spans[0][0]
print(f"dispatches {len(spans)}  span {total / 1e6:.1f} ms  busy {busy / 1e6:.1f} ms ({100 * busy / total:.1f} %)")
for bucket in ("<5us", "5-20us", "20-100us", ">100us"):
    print(f"  idle in gaps {bucket:9s} {gaps[bucket] / 1e6:8.2f} ms")
PY
fi
