# k_overlap on the GPU box: its tests, scripts/overlap_probe.py, the kernel trace of the probe (rocprofv3)
mkdir -p gpurun_out/ov
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "overlap or inner or one_hot or timing" > gpurun_out/ov/tests.log 2>&1; tail -5 gpurun_out/ov/tests.log
for v in ""; do
  echo "== variant ${v:-product}"
  OV_LIB=${v:+$GRAFT_REPO_ROOT/unified_cvo_amd/lib/libcvo_hip_$v.so} timeout 200 python scripts/overlap_probe.py 2>&1 | cut -c1-120
done > gpurun_out/ov/probe.txt 2>&1
cat gpurun_out/ov/probe.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ovp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ovp -o ov -- python $GRAFT_REPO_ROOT/scripts/overlap_probe.py > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/ovp/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "overlap" in r["Name"] or "tile_sph" in r["Name"]: print(r["Name"].split("(")[0][-40:], r["Calls"], "avg %.1f us"%(float(r["AverageNs"])/1e3), "min %.1f max %.1f"%(float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
