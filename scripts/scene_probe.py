import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np
import cases
from unified_cvo_amd import CvoGPU
for n in (3000, 10000):
    P, src, tgt, init = cases.scene(n=n)
    g = CvoGPU(params=P); da, db = g.upload(src), g.upload(tgt)
    g.align(da, db, init, max_iterations=20)
    g.set_option("VERBOSE", "1")
    r = g.align(da, db, init)
    g.set_option("VERBOSE", None)
    print(n, r.iterations, r.ret, f"{r.seconds*1e3:.1f} ms, {r.seconds*1e6/r.iterations:.1f} us/it", g.debug_list_builds(), flush=True)
