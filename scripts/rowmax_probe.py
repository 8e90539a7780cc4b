"""Rows beyond ROW_MAX candidates leave the thread-per-row kernel for k_assoc_dense (a wave per row, long lists): us per
iteration of ONE pair in flight for several limits, and whether the poses stay bit-identical (they must: every row joins
k_assoc's reduction at its own position, whoever evaluated it).  usage: rowmax_probe.py [limit ...]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases
from unified_cvo_amd import CvoGPU

limits = [int(a) for a in sys.argv[1:]] or [64, 24, 16, 12, 8]
work = [("scene10k", cases.scene, dict(n=10000), 600), ("scene3k", cases.scene, dict(n=3000), 600),
        ("demo", cases.config1, {}, 1000), ("geo10k", cases.config2, dict(n=10000), 0),
        ("config3", cases.config3, dict(n=10000), 1500), ("config4", cases.config4, dict(n=10000), 0)]
for name, builder, kw, mi in work:
    P, a, b, init = builder(**kw)
    g = CvoGPU(params=P, library=os.environ.get("CVO_LIB") or None)
    da, db = g.upload(a), g.upload(b)
    ref = None
    line = f"{name:9s}"
    for lim in limits:
        g.set_option("ROW_MAX", str(lim))
        g.align(da, db, init, max_iterations=40)
        best = None
        for _ in range(2):
            r = g.align(da, db, init, max_iterations=mi)
            if best is None or r.seconds < best.seconds:
                best = r
        ovf, scanned, dense = g.debug_row_classes(0)
        same = "" if ref is None or np.array_equal(ref.transform, best.transform) else " DIFFERENT"
        ref = ref or best
        line += f" | {lim:2d}: {best.seconds * 1e6 / max(best.iterations, 1):6.2f} us/it ({ovf} ovf){same}"
    print(line, flush=True)
    g.close()
