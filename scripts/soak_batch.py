"""Soak of the batch scheduler: random batches (2-24 pairs; slab and clustered pairs of random, ragged sizes; random
iteration caps, lengthscales and neighbour caps; every other trial with the on-device list verification) - every pair of
every batch must end bit-identical to the same pair solved alone.  usage: soak_batch.py [trials] [first seed]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases
from unified_cvo_amd import CvoGPU, CvoPointCloud, synth

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
t0 = time.perf_counter()
for trial in range(seed0, seed0 + trials):
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
    verify = trial % 2 == 0 and n_it != 0
    if verify:
        os.environ["CVO_VERIFY_LISTS"] = "1"
    try:
        gpu = CvoGPU(params=P, library=os.environ.get("CVO_LIB") or None)
        res = gpu.align_batch([p[0] for p in pairs], [p[1] for p in pairs], [p[2] for p in pairs], max_iterations=n_it)
    finally:
        os.environ.pop("CVO_VERIFY_LISTS", None)
    solo = CvoGPU(params=P, library=os.environ.get("CVO_LIB") or None)
    diff = []
    for q, (p, r) in enumerate(zip(pairs, res)):
        one = solo.align(p[0], p[1], p[2], max_iterations=n_it)
        if not (np.array_equal(one.transform, r.transform) and (one.iterations, one.ret, one.final_ell, one.final_num_neighbors) ==
                (r.iterations, r.ret, r.final_ell, r.final_num_neighbors)):
            diff.append(q)
    classes = [gpu.debug_row_classes(q) for q in range(n_pairs)]
    print(f"[soak] trial {trial}: {n_pairs} pairs, ell {P.ell_init} K {P.nearest_neighbors_max}, {n_it or 'full'} iterations, verify {int(verify)}, "
          f"pairs with overflow rows {sum(c[0] > 0 for c in classes)}, dense regime {sum(c[2] for c in classes)}: "
          f"{'OK' if not diff else 'DIFFERENT ' + str(diff)}", flush=True)
    bad += bool(diff)
    gpu.close(); solo.close()
print(f"[soak] {trials} trials in {time.perf_counter() - t0:.0f} s, {bad} with differences")
sys.exit(1 if bad else 0)
