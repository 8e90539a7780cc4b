"""Generates tests/golden/micro_ell.npz: kernel-level micro-fixtures of fill_in_A_mat_gpu.

256 x 256 clouds, the FULL ELL matrix (`mat`, `ind_row2col`, `nonzeros`; layout of SparseKernelMat.hpp:11-19, row
stride = K) of ONE association pass - the semantics of /root/reference/src/cvo/CvoGPU.cu:477-593 - for
  geo           geometry only                        (cvo_geometric_params_gpu.yaml)
  geo_colour    geometry x colour                    (cvo_intensity_params_gpu.yaml + documented overrides)
  geo_col_sem   geometry x colour x semantics        (cvo_semantic_params_img_gpu0.yaml, warm start pose)
  kcap          geometry only with K = 6: every row is cut by the ordered first-K truncation
Inputs (points, features, labels, pose, ell, K) are stored next to the outputs, so the fixture is self-contained
DATA: tests compare the oracle (CPU) and the HIP path (GPU) against it without regenerating anything.  The values
are outputs of oracle/ in this container (the reference has no golden vectors and cannot be built here: "parity
unpinned"); they pin both implementations against regressions of the kernel semantics.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from unified_cvo_amd import synth  # noqa: E402

N = 256
# name -> (builder, pose, ell, K).  The clouds are the first 256 points of a 2000-point synthetic pair (so that the
# point density - hence the number of neighbours inside a cut-off radius - is that of the test-sized configurations).
SPECS = {
    "geo": (cases.config2, "identity", 0.6, 64),
    "geo_colour": (cases.config3, "identity", 1.5, 64),
    "geo_col_sem": (cases.config4, "warm", 1.5, 64),
    "kcap": (cases.config2, "identity", 0.6, 6),
}


def build(name):
    builder, pose, ell, K = SPECS[name]
    P, src, tgt, init = builder(n=2000)
    xs, fs, ls, gs = src.device_arrays()
    xt, ft, lt, gt = tgt.device_arrays()
    # 256 points that are neighbours in space (sorted by depth, a slab of the frustum), in the generator's order
    pick_s = np.sort(np.argsort(xs[:, 2], kind="stable")[:N])
    pick_t = np.sort(np.argsort(xt[:, 2], kind="stable")[:N])
    sub = lambda a, idx: None if a is None else np.ascontiguousarray(a[idx])
    X = (sub(xs, pick_s), sub(fs, pick_s), sub(ls, pick_s), sub(gs, pick_s))
    Y = (sub(xt, pick_t), sub(ft, pick_t), sub(lt, pick_t), sub(gt, pick_t))
    T = np.eye(4, dtype=np.float32) if pose == "identity" else (synth.gt_motion() @ synth.warm_start_delta()).astype(np.float32)
    return P, X, Y, T, ell, K


def evaluate(oracle, P, X, Y, T, ell, K):
    o = oracle.iteration(oracle.params_from(P), oracle.Cloud(*X), oracle.Cloud(*Y), T[:3, :3], T[:3, 3], ell, K,
                         want_ell=True)
    return o["mat"], o["ind"], o["nonzeros"]


if __name__ == "__main__":
    po.set_num_threads(4)
    out = {}
    for name in SPECS:
        P, X, Y, T, ell, K = build(name)
        mat, ind, nz = evaluate(po, P, X, Y, T, ell, K)
        print(f"{name}: nnz {int(nz.sum())}, rows on the cap {int((nz == K).sum())} / {N}, max {int(nz.max())}, K {K}")
        assert nz.sum() > 0
        if name == "kcap":
            assert (nz == K).sum() > N // 2
        else:
            assert nz.max() < K   # no truncation: the full pattern is in the fixture
        for key, arr in (("xs", X[0]), ("fs", X[1]), ("ls", X[2]), ("gs", X[3]), ("xt", Y[0]), ("ft", Y[1]), ("lt", Y[2]),
                         ("gt", Y[3])):
            if arr is not None:
                out[f"{name}/{key}"] = arr
        out[f"{name}/T"] = T
        out[f"{name}/ell_K"] = np.array([ell, K], np.float64)
        out[f"{name}/mat"] = mat.astype(np.float32)
        out[f"{name}/ind"] = ind.astype(np.int32)
        out[f"{name}/nonzeros"] = nz.astype(np.uint32)
    path = os.path.join(ROOT, "tests", "golden", "micro_ell.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
