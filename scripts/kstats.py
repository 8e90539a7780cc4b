"""Per-kernel duration statistics from a rocprofv3 kernel_trace.csv, early-exit launches (< cut ns) listed apart."""
import csv, sys, collections
cut = float(sys.argv[2]) if len(sys.argv) > 2 else 4500.0
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r["Kernel_Name"].split("(")[0][-40:]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    big = [x for x in v if x >= cut]
    print(f"{k:42s} n={len(v):6d} tot={sum(v)/1e6:8.2f}ms  work n={len(big):6d} avg={sum(big)/max(1,len(big))/1e3:7.2f}us tot={sum(big)/1e6:8.2f}ms")
