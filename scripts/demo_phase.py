"""Where one pair in flight spends an iteration: the phase stamps of k_assoc / k_coeff and of the update tail (library
stderr) for a named case.  usage: CVO_PHASE_TICKS=1 python scripts/demo_phase.py [demo|scene3k|scene10k|geo10k] [iterations]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("CVO_PHASE_TICKS", "1")
import cases
from unified_cvo_amd import CvoGPU
name = sys.argv[1] if len(sys.argv) > 1 else "demo"
mi = int(sys.argv[2]) if len(sys.argv) > 2 else 300
builder, kw = {"scene10k": (cases.scene, dict(n=10000)), "scene3k": (cases.scene, dict(n=3000)), "demo": (cases.config1, {}),
               "geo10k": (cases.config2, dict(n=10000))}[name]
P, a, b, init = builder(**kw)
g = CvoGPU(params=P)
da, db = g.upload(a), g.upload(b)
r = g.align(da, db, init, max_iterations=mi)
print(name, r.iterations, f"{r.seconds*1e6/max(r.iterations,1):.2f} us/it (instrumented kernels)", file=sys.stderr)
ka, kc = g.debug_time_kernels(20)
print(f"alone on the GPU: k_assoc {ka*1e3:.2f} us, k_coeff {kc*1e3:.2f} us", file=sys.stderr)
