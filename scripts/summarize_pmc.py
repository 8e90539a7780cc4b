"""Summarises rocprofv3 --pmc counter_collection.csv files (per kernel: launches, average per launch) into one JSON.
Usage: summarize_pmc.py OUT.json DIR [DIR ...]"""
import collections
import csv
import glob
import json
import sys

out = {}
for d in sys.argv[2:]:
    for f in glob.glob(d + "/*counter_collection.csv"):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for row in csv.DictReader(open(f)):
            if "cvo_dev" not in row["Kernel_Name"]:
                continue
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            k = (name, row["Counter_Name"])
            agg[k][0] += 1
            agg[k][1] += float(row["Counter_Value"])
        for (name, c), (n, s) in sorted(agg.items()):
            out.setdefault(name, {})[c] = {"launches": n, "avg_per_launch": s / n}
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps({k: {c: float("%.4g" % x["avg_per_launch"]) for c, x in v.items()} for k, v in out.items()}, indent=1))
