"""List-build statistics of the BASELINE configs (one pair each): builds, iterations, candidate evaluations."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from unified_cvo_amd import CvoGPU
for name, b, kw in (("config1", cases.config1, {}), ("config2 5k", cases.config2, dict(n=5000)), ("config2 10k", cases.config2, dict(n=10000)),
                    ("config3", cases.config3, dict(n=10000)), ("config4", cases.config4, dict(n=10000))):
    P, src, tgt, init = b(**kw)
    gpu = CvoGPU(params=P)
    g = gpu.align(src, tgt, init)
    bl, it, ca = gpu.debug_list_builds()
    print(f"{name}: iterations {g.iterations}, {1e6*g.seconds/max(g.iterations,1):.1f} us/iter, builds {bl} ({100.0*bl/max(it,1):.1f}%), candidates/row/iter {ca/max(it,1)/src.num_points():.1f}")
