"""inner_product_gpu / function_angle on resident 10k x 10k clouds: wall time per call (median of 15), configs 2 / 3 / 4,
and the values next to the batch-free reference path (cvo_association's chain sums the same matrix)."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases
from unified_cvo_amd import CvoGPU


def med(fn, reps=15):
    fn(); fn()
    ts = []
    for _ in range(reps):
        t1 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t1)
    return sorted(ts)[len(ts) // 2] * 1e6


for name, builder, kw in (("config2 10k", cases.config2, dict(n=10000)), ("config3 10k", cases.config3, dict(n=10000)),
                          ("config4 10k", cases.config4, dict(n=10000)), ("config2 5k", cases.config2, dict(n=5000))):
    P, a, b, init = builder(**kw)
    g = CvoGPU(params=P, library=os.environ.get("OV_LIB") or None)
    da, db = g.upload(a), g.upload(b)
    ip = med(lambda: g.inner_product_gpu(da, db, init, P.ell_init))
    fa = med(lambda: g.function_angle(da, db, init, P.ell_init, True))
    fe = med(lambda: g.function_angle(da, db, init, P.ell_init, False))
    v = g.inner_product_gpu(da, db, init, P.ell_init)
    g.set_option("IP_CHAIN", "1")   # the list chain (what the loop uses), for comparison
    ipc = med(lambda: g.inner_product_gpu(da, db, init, P.ell_init))
    fec = med(lambda: g.function_angle(da, db, init, P.ell_init, False))
    vc = g.inner_product_gpu(da, db, init, P.ell_init)
    g.set_option("IP_CHAIN", None)
    A = g.compute_association_gpu(da, db, init, P.ell_init)   # (the k_update path: the same matrix, exported)
    chk = f"  sum of the exported matrix {float(np.sum(np.asarray(A[2], np.float64))):.6f}"
    print(f"{name:12s} inner_product_gpu {ip:7.1f} us   function_angle approximate {fa:7.1f} us  exact {fe:7.1f} us   value {v:.6f}{chk}"
          f"   | list chain: {ipc:7.1f} us, exact {fec:7.1f} us, value {vc:.6f}", flush=True)
    g.close()
