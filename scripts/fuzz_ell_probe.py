"""For a seed of tests/test_gpu_parity.py::test_randomised_clustered_trajectories: the full ELL matrix of iteration IT on the
GPU against the oracle's (same pose, ell, K): pattern equal? how many values differ, by how many ulps?"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np, cases
from unified_cvo_amd import CvoGPU, CvoPointCloud, synth
from oracle import pyoracle as po
seed, it = int(sys.argv[1]), int(sys.argv[2])
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
po.set_num_threads(16)
op = po.params_from(P); ox, oy = po.Cloud.from_pointcloud(a), po.Cloud.from_pointcloud(b)
gpu = CvoGPU(params=P)
g = gpu.align(a, b, init, max_iterations=it, trace_capacity=it + 2, trace_dense=it + 2) if it else None
state = np.linalg.inv(g.transform.astype(np.float64)).astype(np.float32) if it else init
ell = g.final_ell if it else P.ell_init
K = g.final_num_neighbors if it else P.nearest_neighbors_max
g1 = gpu.align(a, b, state, max_iterations=1, ell0=ell, K0=K, trace_capacity=2, trace_dense=2)
mat, ind, nz = gpu.debug_last_ell(a.num_points(), K)
o = po.iteration(op, ox, oy, state[:3, :3], state[:3, 3], ell, K, want_ell=True)
print("state after", it, "iterations; ell", ell, "K", K, "nnz", int(nz.sum()), int(o["nonzeros"].sum()))
print("pattern equal:", np.array_equal(nz, o["nonzeros"]) and np.array_equal(ind, o["ind"]))
d = mat.view(np.int32).astype(np.int64) - o["mat"].view(np.int32).astype(np.int64)
valid = np.arange(mat.shape[1])[None, :] < nz[:, None]
print("values differing:", int((d[valid] != 0).sum()), "of", int(valid.sum()), "max ulps", int(np.abs(d[valid]).max()))
t, ot = g1.trace[0], o["trace"]
print("B gpu %.17g oracle %.17g rel %.2e" % (t.B, ot.B, abs(t.B - ot.B) / abs(ot.B)), " sum|terms| unknown; C rel %.2e" % (abs(t.C - ot.C) / abs(ot.C)))
