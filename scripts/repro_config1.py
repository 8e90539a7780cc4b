"""Runs the demo pair (config 1) N times (argv[1], default 8) and prints the iteration counts: the pair stops on an
accidentally small step, so any run-to-run difference in a single bit shows up as a different count (6661 expected).
CVO_NO_DENSE_REGIME=1 / CVO_NO_LEAN=1 exercise the list + overflow path and the full graph."""
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
