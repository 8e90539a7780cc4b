import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from unified_cvo_amd import CvoGPU
P, src, tgt, init = cases.config1()
gpu = CvoGPU(params=P)
its = []
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    g = gpu.align(src, tgt, init)
    its.append(g.iterations)
print("iterations", its)
