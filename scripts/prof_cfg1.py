import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from unified_cvo_amd import CvoGPU
P, src, tgt, init = cases.config1()
gpu = CvoGPU(params=P)
g = gpu.align(src, tgt, init, max_iterations=300)
print(g.iterations, g.seconds)
g = gpu.align(src, tgt, init, max_iterations=1000)
print(g.iterations, g.seconds, "us/iter", 1e6 * g.seconds / g.iterations)
print("builds, iterations, candidates:", gpu.debug_list_builds())
