#!/bin/bash
# per-kernel time of one case: config_profile.sh <case> [n] [its] -> gpurun_out/cfg_prof/<case>_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/cfg_prof
rm -rf /tmp/cp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cp -o c -- python $R/scripts/config_profile.py "$@" > $R/gpurun_out/cfg_prof/$1_run.txt 2>&1
f=$(find /tmp/cp -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/cfg_prof/$1_kernel_stats.csv
grep "^$1\|pairs, " $R/gpurun_out/cfg_prof/$1_run.txt
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r['Name'].split('(')[0].replace("void ","")[:60].ljust(60), r['Calls'].rjust(6), "%10.2f ms"%(float(r['TotalDurationNs'])/1e6), "%9.1f us avg"%(float(r['AverageNs'])/1e3), r['Percentage'])
PY
