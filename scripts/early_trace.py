"""Kernel-trace target for the early phase: one batched align of 64 x 10k x 10k cut after EARLY_ITERS iterations
(run under rocprofv3 --kernel-trace --stats)."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from unified_cvo_amd import CvoGPU
P = cases.load_params("geometric_gpu")
NP = 64
pairs = [cases.config2(n=10000, pair_id=p) for p in range(NP)]
gpu = CvoGPU(params=P)
both = gpu.upload_many([a[1] for a in pairs] + [a[2] for a in pairs])
its = int(os.environ.get("EARLY_ITERS", "64"))
for _ in range(3):
    r = gpu.align_batch(both[:NP], both[NP:], [a[3] for a in pairs], max_iterations=its)
print(its, r[0].seconds, gpu.debug_list_builds())
