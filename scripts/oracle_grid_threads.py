"""Best-effort CPU timing: the oracle's uniform-grid variant at several thread counts (full align() of pair 0, 10k x 10k)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from oracle import pyoracle as po
P, src, tgt, init = cases.config2(n=10000)
op = po.params_from(P); X = po.Cloud.from_pointcloud(src); Y = po.Cloud.from_pointcloud(tgt)
for grid in (1, 0):
    po.set_grid(grid)
    for t in [int(x) for x in os.environ.get("THREADS", "1,4,8,16").split(",")]:
        po.set_num_threads(t)
        po.align(op, X, Y, init, max_iterations=5)
        it = 2000 if grid else 300
        o = po.align(op, X, Y, init, max_iterations=it)
        print(f"grid={grid} threads={t}: {o['seconds']*1e3/o['iterations']:.3f} ms/iteration over {o['iterations']} iterations", flush=True)
po.set_grid(0)
