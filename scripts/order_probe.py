import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from unified_cvo_amd import CvoGPU
# headline batch + real data (config 1 demo) + config 3 single
P = cases.load_params("geometric_gpu")
NP = 64
pairs = [cases.config2(n=10000, pair_id=p) for p in range(NP)]
gpu = CvoGPU(params=P)
both = gpu.upload_many([a[1] for a in pairs] + [a[2] for a in pairs])
inits = [a[3] for a in pairs]
gpu.align_batch(both[:NP], both[NP:], inits, max_iterations=64)
best = 1e9
for _ in range(4):
    t0 = time.perf_counter(); r = gpu.align_batch(both[:NP], both[NP:], inits); best = min(best, time.perf_counter() - t0)
print(f"batch {best*1e3:.2f} ms", end="  ")
for name, b, kw in (("config1", cases.config1, {}), ("config3", cases.config3, dict(n=10000))):
    Pc, a, bb, init = b(**kw)
    g = CvoGPU(params=Pc); da, db = g.upload(a), g.upload(bb)
    g.align(da, db, init, max_iterations=50)
    bt = min(g.align(da, db, init, max_iterations=3000).seconds for _ in range(2))
    print(f"{name} {bt*1e3:.2f} ms", end="  ")
print(os.environ.get("CVO_ORDER", "kd"))
