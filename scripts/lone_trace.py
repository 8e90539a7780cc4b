"""Kernel-trace target: ONE pair in flight (rocprofv3 --kernel-trace --stats -- python scripts/lone_trace.py NAME [iterations])."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from unified_cvo_amd import CvoGPU
name = sys.argv[1]
mi = int(sys.argv[2]) if len(sys.argv) > 2 else 0
builder, kw = {"scene10k": (cases.scene, dict(n=10000)), "scene3k": (cases.scene, dict(n=3000)), "demo": (cases.config1, {}),
               "geo10k": (cases.config2, dict(n=10000)), "config3": (cases.config3, dict(n=10000)),
               "config4": (cases.config4, dict(n=10000))}[name]
P, a, b, init = builder(**kw)
g = CvoGPU(params=P)
da, db = g.upload(a), g.upload(b)
g.align(da, db, init, max_iterations=30)
r = g.align(da, db, init, max_iterations=mi)
print(name, r.iterations, f"{r.seconds*1e6/max(r.iterations,1):.2f} us/it", g.debug_row_classes(0), g.debug_list_builds())
