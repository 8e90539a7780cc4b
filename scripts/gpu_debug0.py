import os, sys, time, faulthandler
faulthandler.enable()
faulthandler.dump_traceback_later(60, exit=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
print("import ok", flush=True)
import cases
from unified_cvo_amd import CvoGPU
p, src, tgt, init = cases.config2(n=512)
print("case ok", flush=True)
gpu = CvoGPU(params=p)
print("ctx ok", flush=True)
d1, d2 = gpu.upload(src), gpu.upload(tgt)
print("upload ok", flush=True)
for ug in (1, 2):
    t0 = time.time()
    g = gpu.align(d1, d2, init, max_iterations=1, trace_capacity=4, trace_dense=4, use_graph=ug)
    print("align 1 iter use_graph=%d ok" % ug, g.iterations, time.time() - t0, flush=True)
    tr = g.trace[0]
    print(tr.k, tr.K, tr.ell, tr.step, tr.nnz, tr.max_nnz, list(tr.omega), list(tr.v), tr.B, tr.C, tr.D, tr.E, tr.dist, flush=True)
t0 = time.time()
g = gpu.align(d1, d2, init, max_iterations=100, use_graph=1)
print("100 iters plain:", time.time() - t0, g.iterations, g.seconds, flush=True)
t0 = time.time()
g = gpu.align(d1, d2, init, max_iterations=100, use_graph=2)
print("100 iters graph:", time.time() - t0, g.iterations, g.seconds, flush=True)
from oracle import pyoracle as po
po.set_num_threads(16)
print("oracle threads", po.num_threads(), flush=True)
op = po.params_from(p)
ox, oy = po.Cloud.from_pointcloud(src), po.Cloud.from_pointcloud(tgt)
t0 = time.time()
o1 = po.iteration(op, ox, oy, init[:3, :3], init[:3, 3], p.ell_init, p.nearest_neighbors_max)
print("oracle iter", time.time() - t0, o1["trace"].nnz, list(o1["trace"].omega), o1["trace"].B, flush=True)
