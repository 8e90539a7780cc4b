"""Batches of clustered 10k scenes: ms, align/s, us per pair-iteration.  usage: scene_batch.py [sizes ...] [OPTION=value ...]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases
from unified_cvo_amd import CvoGPU
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 4, 16, 64]
opts = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
for NP in sizes:
    pairs = [cases.scene(n=10000, pair_id=p) for p in range(NP)]
    P = pairs[0][0]
    gpu = CvoGPU(params=P, library=os.environ.get("CVO_LIB") or None)
    for k, v in opts.items():
        gpu.set_option(k, v)
    both = gpu.upload_many([a[1] for a in pairs] + [a[2] for a in pairs])
    inits = [a[3] for a in pairs]
    gpu.align_batch(both[:NP], both[NP:], inits, max_iterations=64)
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter(); r = gpu.align_batch(both[:NP], both[NP:], inits); best = min(best, time.perf_counter() - t0)
    its = sum(x.iterations for x in r)
    print(f"{NP:3d} clustered 10k pairs: {best*1e3:8.2f} ms, {NP/best:7.1f} align/s, {best*1e6/its*1:6.3f} us per pair-iteration, iterations {r[0].iterations}", flush=True)
    gpu.close()
