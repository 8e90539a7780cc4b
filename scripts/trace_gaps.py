"""Analyses a rocprofv3 kernel_trace.csv: per-queue gaps between consecutive kernels and GPU-wide concurrency."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cvo_dev::", ""), r.get("Queue_Id", "0")) for r in rows if "cvo_dev" in r["Kernel_Name"]]
ks.sort()
t0 = ks[len(ks) // 2][0]
win = [k for k in ks if t0 <= k[0] < t0 + 3_000_000]  # 3 ms window mid-run
byq = collections.defaultdict(list)
for k in win:
    byq[k[3]].append(k)
print("kernels in window:", len(win), "queues:", {q: len(v) for q, v in byq.items()})
gaps = collections.defaultdict(list)
for q, v in byq.items():
    for a, b in zip(v, v[1:]):
        gaps[(a[2][:12], b[2][:12])].append(b[0] - a[1])
for key, g in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:8]:
    g.sort()
    print(f"{key[0]:>14} -> {key[1]:<14} n={len(g):4d} gap median {g[len(g)//2]/1e3:7.2f} us  p10 {g[len(g)//10]/1e3:7.2f}  p90 {g[9*len(g)//10]/1e3:7.2f}")
dur = collections.defaultdict(list)
for k in win:
    dur[k[2][:12]].append(k[1] - k[0])
for n, d in dur.items():
    d.sort()
    print(f"{n:>14} n={len(d):4d} dur median {d[len(d)//2]/1e3:7.2f} us p90 {d[9*len(d)//10]/1e3:7.2f}")
# concurrency: time-weighted number of running kernels
ev = []
for k in win:
    ev.append((k[0], 1)); ev.append((k[1], -1))
ev.sort()
cur = 0; last = ev[0][0]; hist = collections.Counter()
for t, d in ev:
    hist[cur] += t - last; last = t; cur += d
tot = sum(hist.values())
print("concurrency (running kernels : share of time):", {c: round(v / tot, 3) for c, v in sorted(hist.items())})
# one queue's timeline sample
q0 = sorted(byq)[0]
for a in byq[q0][:14]:
    print(f"  q{q0} {a[2][:10]:>10} start {(a[0]-t0)/1e3:8.2f} end {(a[1]-t0)/1e3:8.2f}")
