import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np, cases
from unified_cvo_amd import CvoGPU, CvoPointCloud, synth
from oracle import pyoracle as po
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 214
rs = np.random.default_rng(900 + seed)
n, m = int(rs.integers(1200, 6500)), int(rs.integers(1200, 6500))
src, tgt, _ = synth.scene_pair(n, 50 + seed, m=m)
a, b = CvoPointCloud.from_xyz(src), CvoPointCloud.from_xyz(tgt)
P = cases.load_params("geometric_gpu")
P.ell_init = float(rs.choice([0.3, 0.6, 0.95, 1.4]))
P.nearest_neighbors_max = int(rs.choice([40, 200, 512]))
P.ell_decay_start = int(rs.choice([5, 30]))
P.is_using_range_ell = int(rs.integers(0, 2))
init = (synth.gt_motion() @ synth.warm_start_delta()).astype(np.float32) if rs.integers(0, 2) else np.eye(4, dtype=np.float32)
print(n, m, P.ell_init, P.nearest_neighbors_max, P.ell_decay_start, P.is_using_range_ell, np.allclose(init, np.eye(4)))
po.set_num_threads(16)
op = po.params_from(P); ox, oy = po.Cloud.from_pointcloud(a), po.Cloud.from_pointcloud(b)
gpu = CvoGPU(params=P)
if os.environ.get("PROBE_VARIANTS"):
    o = po.align(op, ox, oy, init, max_iterations=12, trace_capacity=20, trace_dense=20)
    for env in ({}, {"CVO_NO_LONG_LISTS": "1"}, {"CVO_VERIFY_LISTS": "1"}, {"CVO_SKIN": "0"}, {"CVO_NO_DENSE_REGIME": "1"}):
        os.environ.update(env)
        try:
            g = CvoGPU(params=P).align(a, b, init, max_iterations=12, trace_capacity=20, trace_dense=20)
            print(env, ["%.2e" % (abs(x.B - y.B) / max(abs(y.B), 1e-300)) for x, y in zip(g.trace, o["trace"])][:8])
        except Exception as e:
            print(env, "ERROR", e)
        for k in env: del os.environ[k]
    sys.exit(0)
for n_it in (70, 0):
    g = gpu.align(a, b, init, max_iterations=n_it, trace_capacity=120, trace_dense=120)
    o = po.align(op, ox, oy, init, max_iterations=n_it, trace_capacity=120, trace_dense=120)
    print("n_it", n_it, "iterations", g.iterations, o["iterations"], "ret", g.ret, o["ret"], "pose diff", cases.max_abs_diff(g.transform, o["transform"]))
    if n_it:
        pairs = list(zip(g.trace, o["trace"]))
        first = next((i for i, (x, y) in enumerate(pairs) if abs(x.B - y.B) > 1e-12 * max(abs(y.B), 1e-300) or x.nnz != y.nnz), len(pairs))
        print("first difference at iteration", first)
        for x, y in pairs[max(first - 3, 0):first + 8]:
            print(x.k, x.K, y.K, x.nnz, y.nnz, x.max_nnz, y.max_nnz, "%.6f %.6f" % (x.ell, y.ell), "%.6g %.6g" % (x.step, y.step), "B rel %.2e" % (abs(x.B - y.B) / max(abs(y.B), 1e-300)))
gt = np.linalg.inv(synth.gt_motion())
print("vs ground truth: gpu %.2e oracle %.2e" % (cases.max_abs_diff(g.transform, gt), cases.max_abs_diff(o["transform"], gt)))
