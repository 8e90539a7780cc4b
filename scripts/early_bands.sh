# the early phase of the headline batch (64 x 10k x 10k) by iteration band for the CURRENT sources - kernel
# time per band and kernel (rocprofv3 --kernel-trace --stats of scripts/early_trace.py cut at 16 / 64 / 256 iterations, three
# runs each, differences between the cuts), launches, list builds, candidates per row - next to the wall time of the cuts
# without the profiler (scripts/early_sweep.py).  usage (GPU box): bash scripts/early_bands.sh > gpurun_out/r6/early_bands.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for it in 16 64 256; do
  rm -rf /tmp/et$it
  EARLY_ITERS=$it rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/et$it -o e -- python $R/scripts/early_trace.py 2>/dev/null | tail -1 > /tmp/et$it.line
done
python3 - <<'PY'
import csv, glob
cuts = (16, 64, 256)
tot, calls, info = {}, {}, {}
for it in cuts:
    f = glob.glob(f"/tmp/et{it}/**/*kernel_stats.csv", recursive=True)[0]
    tot[it], calls[it] = {}, {}
    for r in csv.DictReader(open(f)):
        nm = r["Name"].split("(")[0].replace("void ", "").replace("cvo_dev::", "").split("<")[0]
        if nm in ("k_kd_order", "k_hold", "k_update"):
            continue
        tot[it][nm] = tot[it].get(nm, 0.0) + float(r["TotalDurationNs"]) / 1e6 / 3    # three runs per cut
        calls[it][nm] = calls[it].get(nm, 0) + int(r["Calls"]) / 3
    line = open(f"/tmp/et{it}.line").read().split()
    info[it] = line
names = sorted({n for it in cuts for n in tot[it]}, key=lambda n: -tot[256].get(n, 0))
print("kernel time per band, ms summed over the four sub-batch streams (divide by 4 for one stream), launches per run in brackets")
print("band      " + "".join(f"{n:>22s}" for n in names) + f"{'sum':>10s}{'sum / 4':>10s}")
prev = None
for it in cuts:
    row, s = "", 0.0
    for n in names:
        t = tot[it].get(n, 0.0) - (tot[prev].get(n, 0.0) if prev else 0.0)
        c = calls[it].get(n, 0) - (calls[prev].get(n, 0) if prev else 0)
        s += t
        row += f"{t:12.2f} [{int(c):6d}]"
    print(f"{(prev or 0):3d}-{it - 1:<5d} " + row + f"{s:10.2f}{s / 4:10.2f}")
    prev = it
print("\n(builds, pair-iterations, candidate evaluations) of one run at each cut:", {it: " ".join(info[it][2:]) for it in cuts})
PY
echo
echo "wall time of the same cuts without the profiler (best of 4):"
cd $R && python scripts/early_sweep.py 2>&1 | tail -1
