# per-kernel time of one pair in flight: lone_profile.sh NAME ITERATIONS  (GPU box)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/lt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lt -o lt -- python $GRAFT_REPO_ROOT/scripts/lone_trace.py $1 $2 2>/dev/null | tail -1
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/lt/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"])>0.8 and "kd_order" not in r["Name"]: print("  ", r["Name"].split("(")[0].replace("void cvo_dev::","")[:50].ljust(50), r["Calls"].rjust(6), "%9.2f ms"%(float(r["TotalDurationNs"])/1e6), "%8.1f us avg"%(float(r["AverageNs"])/1e3))
PY
