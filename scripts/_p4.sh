R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p4
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p4 -o c4 -- python $R/scripts/run_config.py config4 --iterations 3 > /tmp/p4.log 2>&1 < /dev/null
echo rc=$?
grep iteration /tmp/p4.log
mkdir -p $R/gpurun_out/p4
F=$(find /tmp/p4 -name "*kernel_stats.csv" | head -1)
cp $F $R/gpurun_out/p4/kernel_stats.csv
head -40 $F | cut -c1-200
T=$(find /tmp/p4 -name "*kernel_trace.csv" | head -1)
python3 - $T <<'PY'
import csv, sys
spans=[]
for row in csv.DictReader(open(sys.argv[1])):
    spans.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
spans.sort()
busy=0; cur_s,cur_e=spans[0]
for s,e in spans[1:]:
    if s>cur_e: busy+=cur_e-cur_s; cur_s,cur_e=s,e
    else: cur_e=max(cur_e,e)
busy+=cur_e-cur_s
print("dispatches",len(spans),"span ms",(spans[-1][1]-spans[0][0])/1e6,"busy ms",busy/1e6)
PY
