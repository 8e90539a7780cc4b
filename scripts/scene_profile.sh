#!/bin/bash
# per-kernel time of the clustered street-scene workload (scripts/scene_probe.py) -> gpurun_out/scene_prof/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/scene_prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -o scene -- python $R/scripts/scene_probe.py > $R/gpurun_out/scene_prof/run.txt 2>&1
f=$(find /tmp/sp -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/scene_prof/kernel_stats.csv
head -14 "$f" | cut -c1-150
