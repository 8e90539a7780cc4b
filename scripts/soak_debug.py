import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases
from unified_cvo_amd import CvoGPU, CvoPointCloud, synth
trial = int(sys.argv[1]); 
rs = np.random.default_rng(5000 + trial)
P = cases.load_params("geometric_gpu")
P.ell_init = float(rs.choice([0.3, 0.6, 0.95, 1.4]))
P.nearest_neighbors_max = int(rs.choice([40, 200, 512]))
P.ell_decay_start = int(rs.choice([5, 30]))
n_pairs = int(rs.integers(2, 25))
big = rs.integers(0, 4) == 0
pairs = []
for q in range(n_pairs):
    n = int(rs.integers(300, 9000 if big else 3500)); m = int(rs.integers(300, 9000 if big else 3500))
    if rs.integers(0, 2):
        s, t, _ = synth.scene_pair(n, 100 * trial + q, m=m)
    else:
        s, t, _ = synth.geometric_pair(n, 100 * trial + q, m=m)
    init = (synth.gt_motion() @ synth.warm_start_delta()).astype(np.float32) if rs.integers(0, 2) else np.eye(4, dtype=np.float32)
    pairs.append((CvoPointCloud.from_xyz(s), CvoPointCloud.from_xyz(t), init))
n_it = int(rs.choice([40, 150, 400, 0])) if not big else int(rs.choice([40, 150]))
print("trial", trial, "pairs", n_pairs, "n_it", n_it, "sizes", [(p[0].num_points(), p[1].num_points()) for p in pairs])
prev = os.path.join(ROOT, "unified_cvo_amd/lib/libcvo_hip_prev.so")
def run(lib, verify, batch):
    if verify: os.environ["CVO_VERIFY_LISTS"] = "1"
    g = CvoGPU(params=P, library=lib)
    os.environ.pop("CVO_VERIFY_LISTS", None)
    if batch:
        r = g.align_batch([p[0] for p in pairs], [p[1] for p in pairs], [p[2] for p in pairs], max_iterations=n_it)
    else:
        r = [g.align(p[0], p[1], p[2], max_iterations=n_it) for p in pairs]
    g.close()
    return r
ref = run(prev, False, False)
for name, lib, v, b in (("prev batch verify", prev, True, True), ("new solo", None, False, False), ("new solo verify", None, True, False), ("new batch", None, False, True), ("new batch verify", None, True, True), ("new batch verify again", None, True, True), ("new solo again", None, False, False)):
    r = run(lib, v, b)
    d = [q for q in range(n_pairs) if not (np.array_equal(r[q].transform, ref[q].transform) and r[q].iterations == ref[q].iterations)]
    print(f"{name:26s} differs from prev solo at {d}", [ (r[q].iterations, ref[q].iterations, float(np.max(np.abs(r[q].transform-ref[q].transform)))) for q in d])
