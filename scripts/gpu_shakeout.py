"""First-contact GPU script: single-iteration parity, trajectory parity and timing.  Not a test; prints diagnostics."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from unified_cvo_amd import CvoGPU  # noqa: E402


def cmp_trace(a, b, tag):
    fields = ["k", "K", "ell", "step", "nnz", "max_nnz"]
    s = " ".join(f"{f}={getattr(a, f)}/{getattr(b, f)}" for f in fields)
    dom = max(abs(a.omega[i] - b.omega[i]) for i in range(3))
    dv = max(abs(a.v[i] - b.v[i]) for i in range(3))
    rel = lambda x, y: abs(x - y) / max(abs(y), 1e-300)
    print(f"  [{tag}] {s} d_omega={dom:.2e} d_v={dv:.2e} relB={rel(a.B, b.B):.1e} relC={rel(a.C, b.C):.1e} "
          f"relD={rel(a.D, b.D):.1e} relE={rel(a.E, b.E):.1e} dist={a.dist:.6e}/{b.dist:.6e} "
          f"dR={max(abs(a.R[i]-b.R[i]) for i in range(9)):.1e} dT={max(abs(a.T[i]-b.T[i]) for i in range(3)):.1e}")


def run_case(name, builder, n_iter_cmp=30, full=True, **kw):
    print(f"=== {name}")
    p, src, tgt, init = builder(**kw)
    gpu = CvoGPU(params=p)
    op = po.params_from(p)
    ox, oy = po.Cloud.from_pointcloud(src), po.Cloud.from_pointcloud(tgt)
    dsrc, dtgt = gpu.upload(src), gpu.upload(tgt)
    # single iteration on shared initial state + ELL
    K = p.nearest_neighbors_max
    g1 = gpu.align(dsrc, dtgt, init, max_iterations=1, trace_capacity=4, trace_dense=4)
    mat, ind, nz = gpu.debug_last_ell(src.num_points(), K)
    ncand = gpu.debug_last_candidates()
    o1 = po.iteration(op, ox, oy, init[:3, :3], init[:3, 3], p.ell_init, K, want_ell=True)
    print("  ELL nonzeros equal:", np.array_equal(nz, o1["nonzeros"]), "ind equal:", np.array_equal(ind, o1["ind"]),
          "mat equal:", np.array_equal(mat, o1["mat"]), "max|dmat|:", float(np.max(np.abs(mat - o1["mat"]))),
          "nnz:", int(nz.sum()), "cand:", ncand)
    if g1.trace:
        cmp_trace(g1.trace[0], o1["trace"], "iter0")
    # trajectory prefix
    t0 = time.time()
    g = gpu.align(dsrc, dtgt, init, max_iterations=n_iter_cmp, trace_capacity=n_iter_cmp, trace_dense=n_iter_cmp)
    tg = time.time() - t0
    o = po.align(op, ox, oy, init, trace_capacity=n_iter_cmp, trace_dense=n_iter_cmp, max_iterations=n_iter_cmp)
    print(f"  prefix {n_iter_cmp} iters: gpu {g.iterations} it {tg*1e3:.1f} ms (loop {g.seconds*1e3:.2f} ms) ; oracle {o['iterations']} it {o['seconds']*1e3:.1f} ms")
    for i in sorted(set([0, 1, 2, n_iter_cmp // 2, n_iter_cmp - 1])):
        if i < len(g.trace) and i < len(o["trace"]):
            cmp_trace(g.trace[i], o["trace"][i], f"k={i}")
    print("  prefix pose max|d|:", cases.max_abs_diff(g.transform, o["transform"]))
    if full:
        t0 = time.time()
        g = gpu.align(dsrc, dtgt, init)
        tg = time.time() - t0
        print(f"  FULL gpu: ret={g.ret} iters={g.iterations} wall={tg*1e3:.1f} ms loop={g.seconds*1e3:.1f} ms "
              f"-> {g.seconds*1e6/max(g.iterations,1):.1f} us/iter, final ell={g.final_ell:.4f} K={g.final_num_neighbors}")
        t0 = time.time()
        g2 = gpu.align(dsrc, dtgt, init)
        print(f"  FULL gpu (2nd): loop={g2.seconds*1e3:.1f} ms same result: {np.array_equal(g.transform, g2.transform)}")
        scan_ms = gpu.debug_time_scan(50)
        print(f"  scan kernel: {scan_ms*1e3:.2f} us/launch  ({src.num_points()*tgt.num_points()/scan_ms/1e6:.1f} Gpairs/s)")
        return gpu, g
    return gpu, g


if __name__ == "__main__":
    which = sys.argv[1:] or ["c2_1k", "c2_5k", "c2_10k", "c4_2k", "c3_2k", "c1"]
    po.set_num_threads(int(os.environ.get("ORACLE_THREADS", "32")))
    print("oracle threads:", po.num_threads(), flush=True)
    if "c2_1k" in which:
        gpu, g = run_case("config2 n=1000", cases.config2, n=1000)
        o = po.align(po.params_from(gpu.params), *[po.Cloud.from_pointcloud(c) for c in cases.config2(n=1000)[1:3]], np.eye(4))
        print("  FULL oracle: iters", o["iterations"], "sec", o["seconds"], "pose max|d| vs gpu:", cases.max_abs_diff(g.transform, o["transform"]))
    if "c2_5k" in which:
        run_case("config2 n=5000", cases.config2, n=5000)
    if "c2_10k" in which:
        run_case("config2 n=10000", cases.config2, n=10000, n_iter_cmp=10)
    if "c4_2k" in which:
        run_case("config4 n=2000 (semantic)", cases.config4, n=2000)
    if "c3_2k" in which:
        run_case("config3 n=2000 (colour)", cases.config3, n=2000)
    if "c1" in which:
        run_case("config1 demo", cases.config1, n_iter_cmp=50, full=False)
    if "batch" in which:
        for nb in (1, 4, 16):
            p, _, _, _ = cases.config2(n=10000)
            gpu = CvoGPU(params=p)
            pairs = [cases.config2(n=10000, pair_id=i) for i in range(nb)]
            src = [gpu.upload(q[1]) for q in pairs]
            tgt = [gpu.upload(q[2]) for q in pairs]
            inits = [q[3] for q in pairs]
            t0 = time.time()
            res = gpu.align_batch(src, tgt, inits)
            dt = time.time() - t0
            print(f"batch {nb} x 10k: wall {dt:.3f} s -> {nb/dt:.2f} align/s ; loop {res[0].seconds:.3f} s ; iters {[r.iterations for r in res][:4]}")
            print("   scan us:", gpu.debug_time_scan(20) * 1e3)
