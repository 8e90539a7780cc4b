"""Per-kernel replay timings (cvo_debug_time_kernels: the per-iteration kernels re-run on the state the last call left
behind) of the product library next to experiment builds, interleaved.  usage: kernel_ab.py [name ...]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases
from unified_cvo_amd import CvoGPU, build as B
names = ["product"] + sys.argv[1:]
NP = int(os.environ.get("EXP_PAIRS", "64"))
ITS = int(os.environ.get("EXP_ITERS", "600"))
P = cases.load_params("geometric_gpu")
pairs = [cases.config2(n=10000, pair_id=p) for p in range(NP)]
ctx = {}
for nm in names:
    lib = None if nm == "product" else os.path.join(B.LIBDIR, f"libcvo_hip_{nm}.so")
    gpu = CvoGPU(params=P, library=lib)
    both = gpu.upload_many([a[1] for a in pairs] + [a[2] for a in pairs])
    gpu.align_batch(both[:NP], both[NP:], [a[3] for a in pairs], max_iterations=ITS)
    ctx[nm] = gpu
for rnd in range(3):
    for nm in names:
        print(nm.ljust(10), ctx[nm].debug_time_kernels(reps=200), flush=True)
