import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import cases
from unified_cvo_amd import CvoGPU
P = cases.load_params("geometric_gpu")
pairs = [cases.config2(n=10000, pair_id=p) for p in range(4)]
gpu = CvoGPU(params=P)
both = gpu.upload_many([a[1] for a in pairs] + [a[2] for a in pairs])
gpu.set_option("VERBOSE", "2")
for it in (300, 500, 700, 900, 1100, 1500, 1900):
    gpu.align_batch(both[:4], both[4:], [a[3] for a in pairs], max_iterations=it)
