#!/usr/bin/env python
"""Run-to-run reproducibility under load (MI355X): the headline batch and the single-pair configs are solved
REPS times each and every returned pose, iteration count and return code must be bit-identical to the first run.
A cross-block ordering bug (a partial read before it was stored, a counter that is not reset) shows up here as a
pose that differs in the last bits once in a few hundred runs, long before a parity tolerance notices.

usage: stress_repeat.py [reps_batch=200] [reps_single=100]
"""
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import cases  # noqa: E402
from unified_cvo_amd import CvoGPU  # noqa: E402

reps_batch = int(sys.argv[1]) if len(sys.argv) > 1 else 200
reps_single = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0


def key(results):
    return [(r.ret, int(r.iterations), r.transform.tobytes()) for r in results]


# ---- headline batch: 64 x 10k x 10k geometric
NP = 64
pairs = [cases.config2(n=10000, pair_id=p) for p in range(NP)]
P = pairs[0][0]
gpu = CvoGPU(params=P)
both = gpu.upload_many([a[1] for a in pairs] + [a[2] for a in pairs])
inits = [a[3] for a in pairs]
t0 = time.perf_counter()
ref = None
for r in range(reps_batch):
    k = key(gpu.align_batch(both[:NP], both[NP:], inits))
    if ref is None:
        ref = k
    elif k != ref:
        bad += 1
        d = [p for p in range(NP) if k[p] != ref[p]]
        print(f"[stress] batch run {r}: pairs {d} differ from run 0", flush=True)
print(f"[stress] batch of {NP}: {reps_batch} runs in {time.perf_counter() - t0:.1f} s, {bad} differing", flush=True)
del gpu, both

# ---- one pair in flight, every config shape (interleaved so that graphs / workspaces are re-keyed between calls)
singles = [("config2 10k", cases.config2(n=10000)), ("config3", cases.config3()), ("config4", cases.config4()),
           ("config1", cases.config1()), ("clustered scene 10k", cases.scene(n=10000))]
for name, c in singles:
    g = CvoGPU(params=c[0])
    s, t = g.upload_many([c[1], c[2]])
    ref = None
    nb = 0
    t0 = time.perf_counter()
    for r in range(reps_single):
        k = key([g.align(s, t, c[3])])
        if ref is None:
            ref = k
        elif k != ref:
            nb += 1
            print(f"[stress] {name} run {r} differs from run 0", flush=True)
    bad += nb
    print(f"[stress] {name}: {reps_single} runs in {time.perf_counter() - t0:.1f} s ({ref[0][1]} iterations), {nb} differing", flush=True)
print("[stress] " + ("OK: every run bit-identical" if bad == 0 else f"FAILED: {bad} differing runs"), flush=True)
sys.exit(1 if bad else 0)
