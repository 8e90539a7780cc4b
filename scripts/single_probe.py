import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from unified_cvo_amd import CvoGPU
for name, b in (("config3", cases.config3), ("config2", cases.config2), ("config4", cases.config4)):
    P, s, t, init = b(n=10000)
    g = CvoGPU(params=P)
    ds, dt = g.upload(s), g.upload(t)
    g.align(ds, dt, init, max_iterations=50)
    r = min((g.align(ds, dt, init) for _ in range(2)), key=lambda r: r.seconds)
    print(name, r.iterations, round(r.seconds * 1e6 / r.iterations, 2), "us/it", g.debug_list_builds(), flush=True)
