mkdir -p gpurun_out/r6
python bench.py > gpurun_out/r6/bench_base.json 2> gpurun_out/r6/bench_base.err
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for it in 16 64 256; do
rm -rf /tmp/et
EARLY_ITERS=$it rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/et -o e -- python $R/scripts/early_trace.py 2>/dev/null | tail -1
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/et/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"])>0.3: print("  ", r["Name"].split("(")[0][:55].ljust(55), r["Calls"].rjust(6), "%8.2f ms"%(float(r["TotalDurationNs"])/1e6/3), "%8.1f us avg"%(float(r["AverageNs"])/1e3), r["Percentage"])
PY
done > $R/gpurun_out/r6/early_base.txt 2>&1
