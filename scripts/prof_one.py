"""Runs one config-2 align (n from argv) for profiling under rocprofv3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from unified_cvo_amd import CvoGPU
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 1
p, src, tgt, init = cases.config2(n=n)
gpu = CvoGPU(params=p)
pairs = [cases.config2(n=n, pair_id=i) for i in range(nb)]
s = [gpu.upload(q[1]) for q in pairs]; t = [gpu.upload(q[2]) for q in pairs]
res = gpu.align_batch(s, t, [q[3] for q in pairs], max_iterations=iters)
print("iters", res[0].iterations, "loop s", res[0].seconds, "us/iter", res[0].seconds * 1e6 / res[0].iterations)
