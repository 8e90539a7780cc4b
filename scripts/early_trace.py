"""Kernel durations of one align_batch as a function of the launch order (rocprofv3 kernel_trace.csv of scripts/early_prof.py)."""
import csv, sys, collections
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "cvo_dev" in r["Kernel_Name"]]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
# last repetition only: split by big gaps
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = np.array([int(r["Start_Timestamp"]) for r in rows])
gaps = np.where(np.diff(starts) > 3e6)[0]
seg = rows[gaps[-1] + 1:] if len(gaps) else rows
base = int(seg[0]["Start_Timestamp"])
byk = collections.defaultdict(list)
for r in seg:
    name = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
    byk[name].append(((int(r["Start_Timestamp"]) - base) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
edges = [0, 2, 4, 6, 9, 12, 16, 20, 25, 30, 40, 60, 100]
print("window(ms)      " + "".join(f"{n:>22s}" for n in ("k_assoc", "k_coeff", "k_assoc_dense", "k_prep", "k_scan", "k_list")))
for a, b in zip(edges[:-1], edges[1:]):
    line = f"{a:3d}-{b:3d}        "
    for n in ("k_assoc", "k_coeff", "k_assoc_dense", "k_prep", "k_scan", "k_list"):
        v = [d for (t, d) in byk[n] if a <= t < b]
        line += f"   n={len(v):4d} avg={np.mean(v) if v else 0:6.1f}us"
    print(line)
