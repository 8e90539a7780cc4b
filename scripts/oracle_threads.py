import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases
from oracle import pyoracle as po
nt = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
po.set_num_threads(nt)
p, src, tgt, init = cases.config2(n=n)
op = po.params_from(p)
ox, oy = po.Cloud.from_pointcloud(src), po.Cloud.from_pointcloud(tgt)
t0 = time.time()
o = po.align(op, ox, oy, init, max_iterations=10)
print(f"threads={nt} n={n}: 10 iters {o['seconds']:.3f}s -> {o['seconds']/10*1e3:.1f} ms/iter -> {o['seconds']/10*2000:.1f} s/align", flush=True)
