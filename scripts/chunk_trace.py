import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from unified_cvo_amd import CvoGPU
P = cases.load_params("geometric_gpu")
NP = 64
pairs = [cases.config2(n=10000, pair_id=p) for p in range(NP)]
gpu = CvoGPU(params=P)
both = gpu.upload_many([a[1] for a in pairs] + [a[2] for a in pairs])
r = gpu.align_batch(both[:NP], both[NP:], [a[3] for a in pairs])
gpu.set_option("VERBOSE", "3")
r = gpu.align_batch(both[:NP], both[NP:], [a[3] for a in pairs])
gpu.set_option("VERBOSE", None)
# per-iteration trace of pair 0: ell, K, nnz, step
r = gpu.align(both[0], both[NP], pairs[0][3], trace_capacity=2000, trace_dense=2000)
for t in r.trace[:300:4]:
    print(t.k, round(t.ell,4), t.K, t.nnz, t.max_nnz, round(t.step,5))
