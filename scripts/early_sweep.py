"""The first iterations of the headline batch under list-reuse settings (options "NAME=value,NAME=value" per argument):
ms for the first 64 / 256 iterations and the whole run, list builds, whether the poses stay the same.
usage: [SWEEP_CASE=config2|config3|config4|scene] [SWEEP_PAIRS=64] early_sweep.py [SKIN_MAX=0.4 SKIN_MAX=0.4,SKIN=1.5 ...]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases
from unified_cvo_amd import CvoGPU
NP = int(os.environ.get("SWEEP_PAIRS", "64"))
builder = {"config2": cases.config2, "config3": cases.config3, "config4": cases.config4, "scene": cases.scene}[os.environ.get("SWEEP_CASE", "config2")]
pairs = [builder(n=10000, pair_id=p) for p in range(NP)]
P = pairs[0][0]
gpu = CvoGPU(params=P, library=os.environ.get("CVO_LIB") or None)
both = gpu.upload_many([a[1] for a in pairs] + [a[2] for a in pairs])
inits = [a[3] for a in pairs]
ref = None
for setting in [""] + sys.argv[1:]:
    opts = dict(kv.split("=") for kv in setting.split(",") if kv)
    for k, v in opts.items():
        gpu.set_option(k, v)
    line = f"{setting or 'default':40s}"
    for its in (64, 256, 0):
        gpu.align_batch(both[:NP], both[NP:], inits, max_iterations=its)
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter()
            r = gpu.align_batch(both[:NP], both[NP:], inits, max_iterations=its)
            best = min(best, time.perf_counter() - t0)
        b, it, c = gpu.debug_list_builds()
        line += f" | {its or 2000:4d} its {best * 1e3:6.2f} ms, {b / NP:5.1f} builds, {c / max(it, 1) / 1e4:5.2f} cand/row/it"
    sig = np.concatenate([x.transform.ravel() for x in r])
    ref = sig if ref is None else ref
    print(line, "" if np.array_equal(sig, ref) else " POSES DIFFER", flush=True)
    for k in opts:
        gpu.set_option(k, None)
