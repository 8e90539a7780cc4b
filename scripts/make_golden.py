"""Generates tests/golden/oracle_traces.json with the oracle in THIS container.

No reference golden vectors exist and the reference cannot be built here, so
these fixtures pin the ORACLE (and, through tests/test_gpu_parity.py, the HIP path) against
regressions; they are not outputs of the reference.  Seeds / sizes are in tests/cases.py."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from unified_cvo_amd import synth  # noqa: E402


def trace_rows(tr):
    return [dict(k=t.k, K=t.K, ell=float(t.ell), step=float(t.step), nnz=int(t.nnz), max_nnz=int(t.max_nnz),
                 omega=[float(x) for x in t.omega], v=[float(x) for x in t.v], B=t.B, C=t.C, D=t.D, E=t.E,
                 dist=t.dist) for t in tr]


def run(name, builder, dense=50, every=100, max_iterations=0, ip_ell=None, **kw):
    P, src, tgt, init = builder(**kw)
    op = po.params_from(P)
    x, y = po.Cloud.from_pointcloud(src), po.Cloud.from_pointcloud(tgt)
    r = po.align(op, x, y, init, trace_capacity=400, trace_dense=dense, trace_every=every,
                 max_iterations=max_iterations)
    ell = P.ell_init if ip_ell is None else ip_ell
    final = np.linalg.inv(r["transform"].astype(np.float64)).astype(np.float32)  # state (R,T) = inverse of output
    out = dict(name=name, kwargs=kw, max_iterations=max_iterations, ret=r["ret"], iterations=r["iterations"],
               transform=r["transform"].astype(np.float64).tolist(), trace=trace_rows(r["trace"]),
               inner_product_init=float(po.inner_product(op, x, y, init, ell)),
               inner_product_final=float(po.inner_product(op, x, y, final, ell)),
               function_angle_init=float(po.function_angle(op, x, y, init, ell, True)),
               function_angle_final_exact=float(po.function_angle(op, x, y, final, ell, False)))
    print(name, "iterations", out["iterations"], "ret", out["ret"], "ip", out["inner_product_init"],
          out["inner_product_final"], "angle", out["function_angle_final_exact"])
    return out


if __name__ == "__main__":
    po.set_num_threads(8)
    cases_out = [
        run("config1_demo_geometric_k1000", cases.config1, dense=50, every=100, max_iterations=1000),
        run("config2_n2000", cases.config2, n=2000),
        run("config3_n2000", cases.config3, n=2000),
        run("config4_n2000", cases.config4, n=2000),
        # not a BASELINE config: a clustered street scene (rows on the lists, beyond them and on the K cap at once)
        run("scene_n2500", cases.scene, n=2500),
        # ... and at the size of the BASELINE shapes: 10k x 10k, every one of its first 300 iterations (thousands of rows
        # beyond their lists, hundreds on the K cap, list rebuilds, the row-class switches of round 5)
        run("scene_n10000_k300", cases.scene, dense=300, every=100, max_iterations=300, n=10000),
    ]
    path = os.path.join(ROOT, "tests", "golden", "oracle_traces.json")
    with open(path, "w") as f:
        json.dump(dict(generator="scripts/make_golden.py", note="oracle-generated; parity unpinned vs the reference",
                       gt_inverse=np.linalg.inv(synth.gt_motion()).tolist(), cases=cases_out), f, indent=1)
    print("wrote", path, os.path.getsize(path), "bytes")
