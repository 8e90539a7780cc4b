"""The headline batch limited to argv[1] iterations, three times (for rocprofv3 --kernel-trace; see early_trace.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from unified_cvo_amd import CvoGPU
P = cases.load_params("geometric_gpu")
pairs = [cases.config2(n=10000, pair_id=p) for p in range(64)]
gpu = CvoGPU(params=P)
both = gpu.upload_many([a[1] for a in pairs] + [a[2] for a in pairs])
src, tgt = both[:64], both[64:]
inits = [a[3] for a in pairs]
it = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for rep in range(3):
    t0 = time.time(); res = gpu.align_batch(src, tgt, inits, max_iterations=it); t1 = time.time()
    print(it, "iterations:", round((t1 - t0) * 1e3, 2), "ms")
