"""One line per setting of the list-reuse knobs (CVO_SKIN / CVO_LEAN_SKIN / CVO_HORIZON_MARGIN from the environment):
headline batch ms, single-pair us per iteration for configs 2 / 3 / 4 at 10k and config 1 (demo), builds, waits."""
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
gpu.align_batch(both[:NP], both[NP:], inits, max_iterations=64)
best = 1e9
for _ in range(4):
    t0 = time.perf_counter(); gpu.align_batch(both[:NP], both[NP:], inits); best = min(best, time.perf_counter() - t0)
b, it, c = gpu.debug_list_builds()
out = f"batch {best*1e3:6.2f} ms builds/pair {b/NP:5.1f} cand/row/it {c/max(it,1)/1e4:5.2f} |"
gpu.close()
for name, bld, kw in (("c2", cases.config2, dict(n=10000)), ("c3", cases.config3, dict(n=10000)), ("c4", cases.config4, dict(n=10000)), ("c1", cases.config1, {})):
    Pc, a, bb, init = bld(**kw)
    g = CvoGPU(params=Pc); da, db = g.upload(a), g.upload(bb)
    g.align(da, db, init, max_iterations=50)
    r = min((g.align(da, db, init, max_iterations=3000 if name == "c1" else 0) for _ in range(2)), key=lambda r: r.seconds)
    out += f" {name} {r.seconds*1e6/r.iterations:6.2f}"
    g.close()
print(out, " ".join(f"{k[4:]}={v}" for k, v in sorted(os.environ.items()) if k.startswith("CVO_")), flush=True)
