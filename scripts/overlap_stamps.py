"""Where the waves of k_overlap spend their time: an experiment build with cycle-counter stamps (-DCVO_OV_STAMPS), one
inner product per config, per-wave phase times (average / max over the waves of the launch).  GPU box only."""
import ctypes as C
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases
from unified_cvo_amd import CvoGPU, build as B

lib = os.path.join(B.LIBDIR, "libcvo_hip_ovstamps.so")
if not os.path.exists(lib):
    B.build_variant("ovstamps", ["-DCVO_OV_STAMPS"], verbose=False)
for name, builder, kw in (("config2 10k", cases.config2, dict(n=10000)), ("config3 10k", cases.config3, dict(n=10000)),
                          ("config4 10k", cases.config4, dict(n=10000))):
    P, a, b, init = builder(**kw)
    g = CvoGPU(params=P, library=lib)
    da, db = g.upload(a), g.upload(b)
    for _ in range(3):
        g.inner_product_gpu(da, db, init, P.ell_init)
    buf = np.zeros((4096, 8), np.uint64)
    assert g.L.cvo_debug_overlap_ticks(buf.ctypes.data_as(C.c_void_p)) == 0
    nw = ((10000 + 63) // 64) * int(os.environ.get("OV_WAVES", "8"))
    t = buf[:nw].astype(np.float64)
    cnt = buf[:nw, 5]
    seen, scanned, flushes = cnt & 0xffff, (cnt >> 16) & 0xffff, cnt >> 32
    f = lambda c: f"{t[:, c].mean():8.0f} / {t[:, c].max():8.0f}"
    print(f"{name}: waves {nw}; ticks avg / max: prologue {f(0)}, cull {f(1)}, scan {f(2)}, evaluations {f(3)}, tail {f(4)}, "
          f"whole wave {f(6)}; tiles seen {seen.mean():.1f} / {seen.max()}, scanned {scanned.mean():.1f} / {scanned.max()}, "
          f"flushes {flushes.mean():.1f} / {flushes.max()}", flush=True)
    g.close()
