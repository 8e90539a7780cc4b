#!/usr/bin/env python
"""Report behind tests/test_oracle_conventions.py: per BASELINE configuration and floating-point convention of the
oracle (oracle/Makefile `variants`), the iteration count, the first iteration whose integer decisions differ from the
default convention's and the final-pose difference.  Writes profiles/r2/convention_sweep.json.  CPU only."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle as po  # noqa: E402
import test_oracle_conventions as toc  # noqa: E402

po.set_num_threads(int(os.environ.get("OMP_NUM_THREADS", "8")))
report = {}
for name, (builder, kw, max_it, tol, n_same) in sorted(toc.CASES.items()):
    P, src, tgt, init = builder(**kw)
    summary, spread = toc.summarize(toc.sweep(po, P, src, tgt, init, max_it))
    report[name] = dict(pose_spread_max_abs=spread, tolerance=tol, conventions=summary)
    print(name, "spread %.3g (tol %.0e)" % (spread, tol), {k: (v["iterations"], v["first_decision_divergence"]) for k, v in summary.items()})
out = os.path.join(ROOT, "profiles", "r2", "convention_sweep.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(report, open(out, "w"), indent=1)
print("wrote", out)
