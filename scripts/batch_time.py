"""Time of the headline batch (64 x 10k x 10k geometric, 2000 iterations) under the current environment switches:
one line, for parameter sweeps.  usage: [CVO_SKIN=.. CVO_SKIN_MAX=.. ...] python scripts/batch_time.py [iterations]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from unified_cvo_amd import CvoGPU
P = cases.load_params("geometric_gpu")
NP = 64
pairs = [cases.config2(n=10000, pair_id=p) for p in range(NP)]
gpu = CvoGPU(params=P)
both = gpu.upload_many([a[1] for a in pairs] + [a[2] for a in pairs])
inits = [a[3] for a in pairs]
its = int(sys.argv[1]) if len(sys.argv) > 1 else 0
gpu.align_batch(both[:NP], both[NP:], inits, max_iterations=64)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); r = gpu.align_batch(both[:NP], both[NP:], inits, max_iterations=its); best = min(best, time.perf_counter() - t0)
b, it, c = gpu.debug_list_builds()
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("CVO_"))
print(f"{best*1e3:8.2f} ms  builds/pair {b/NP:6.1f}  cand/row/iter {c/max(it,1)/1e4:6.2f}   {tag}", flush=True)
