import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from unified_cvo_amd import CvoGPU
NP = int(os.environ.get("NP", "16"))
pairs = [cases.scene(n=10000, pair_id=p) for p in range(NP)]
gpu = CvoGPU(params=pairs[0][0])
both = gpu.upload_many([a[1] for a in pairs] + [a[2] for a in pairs])
r = gpu.align_batch(both[:NP], both[NP:], [a[3] for a in pairs])
print(r[0].iterations, r[0].seconds, gpu.debug_row_classes(0), gpu.debug_list_builds())
