"""One pair in flight under context options: us per iteration (best of 5), chunks / list builds / waits.
usage: single_sweep.py CASE [OPTION=value[,OPTION=value] ...]   CASE = config2 | config3 | config4 | scene | demo"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases
from unified_cvo_amd import CvoGPU
name = sys.argv[1]
builder, kw, mi = {"config2": (cases.config2, dict(n=10000), 0), "config3": (cases.config3, dict(n=10000), 0),
                   "config4": (cases.config4, dict(n=10000), 0), "scene": (cases.scene, dict(n=10000), 0),
                   "demo": (cases.config1, {}, 1000)}[name]
P, a, b, init = builder(**kw)
g = CvoGPU(params=P)
da, db = g.upload(a), g.upload(b)
ref = None
for setting in [""] + sys.argv[2:]:
    opts = dict(kv.split("=") for kv in setting.split(",") if kv)
    for k, v in opts.items():
        g.set_option(k, v)
    g.align(da, db, init, max_iterations=40)
    best = min((g.align(da, db, init, max_iterations=mi) for _ in range(5)), key=lambda r: r.seconds)
    builds = g.debug_list_builds()[0]
    same = "" if ref is None or np.array_equal(ref, best.transform) else "  POSE DIFFERS"
    ref = best.transform if ref is None else ref
    print(f"{name} {setting or 'default':32s} {best.iterations:5d} iterations {best.seconds * 1e6 / max(best.iterations, 1):7.2f} us/it  {best.seconds * 1e3:7.3f} ms  builds {builds}{same}", flush=True)
    for k in opts:
        g.set_option(k, None)
