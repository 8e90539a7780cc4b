#!/usr/bin/env python
"""Report behind tests/test_oracle_conventions.py: per BASELINE configuration and floating-point convention of the
oracle (oracle/Makefile `variants`), the iteration count, the first iteration whose integer decisions differ from the
default convention's and the final-pose difference.  Writes profiles/r3/convention_sweep.json (test-sized cases) or,
with --full, profiles/r3/convention_sweep_full.json (the BASELINE.json configurations at their literal sizes: config 2
at 5k x 5k, its 10k shape = config 5's per-pair shape, configs 3 and 4 at 10k x 10k).  CPU only."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle as po  # noqa: E402
import test_oracle_conventions as toc  # noqa: E402

po.set_num_threads(int(os.environ.get("OMP_NUM_THREADS", "8")))
FULL = "--full" in sys.argv
CASES = toc.CASES_FULL if FULL else toc.CASES
report = {}
for name, (builder, kw, max_it, tol, n_same) in sorted(CASES.items()):
    P, src, tgt, init = builder(**kw)
    summary, spread = toc.summarize(toc.sweep(po, P, src, tgt, init, max_it))
    report[name] = dict(pose_spread_max_abs=spread, tolerance=tol, conventions=summary)
    print(name, "spread %.3g (tol %.0e)" % (spread, tol), {k: (v["iterations"], v["first_decision_divergence"]) for k, v in summary.items()})
out = os.path.join(ROOT, "profiles", "r3", "convention_sweep_full.json" if FULL else "convention_sweep.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(report, open(out, "w"), indent=1)
print("wrote", out)
