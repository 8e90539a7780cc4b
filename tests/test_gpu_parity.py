"""GPU parity tests proper: the HIP path (through the C-ABI) against the oracle on the same seeded inputs.

Layers (hard part 6):
  (i)   single-iteration parity on shared state: integer/index work (ELL pattern, nnz, max_nnz, K) must be
        bit-exact; floating point within the tolerances written next to each assert;
  (ii)  trajectory-prefix parity at fixed iteration counts;
  (iii) final-pose parity: 1e-4 (max-abs over the 4x4) where termination is well conditioned, 2e-4
        (= 2 * min_step) for runs that end clamped at min_step (configs 2/3).
"""
import json
import os

import numpy as np
import pytest

import cases
from unified_cvo_amd import CvoGPU, CvoPointCloud, CvoParams, CvoError, synth

pytestmark = pytest.mark.gpu

TOL_TWIST = 1e-5      # on the normalised twist (unit 6-vector)
TOL_COEF_REL = 1e-9   # B, C, D, E are double sums of identical float terms; only the order differs
TOL_POSE = 1e-4
TOL_POSE_CLAMPED = 2e-4
TOL_IP_REL = 1e-4


def _ocloud(oracle, pc):
    return oracle.Cloud.from_pointcloud(pc)


def _cmp_trace(a, b):
    assert (a.k, a.K, a.nnz, a.max_nnz) == (b.k, b.K, b.nnz, b.max_nnz), (a.k, a.K, b.K, a.nnz, b.nnz)
    assert a.ell == b.ell
    assert np.allclose(list(a.omega) + list(a.v), list(b.omega) + list(b.v), rtol=0, atol=TOL_TWIST), a.k
    for n in "BCDE":
        x, y = getattr(a, n), getattr(b, n)
        assert abs(x - y) <= TOL_COEF_REL * max(abs(y), 1e-12) + 1e-15, (a.k, n, x, y)
    assert a.step == pytest.approx(b.step, rel=1e-6)
    assert a.dist == pytest.approx(b.dist, rel=1e-5, abs=1e-12)
    assert np.allclose(list(a.R) + list(a.T), list(b.R) + list(b.T), rtol=0, atol=1e-6)


def _single_iteration(oracle, P, src, tgt, init, ell=None, K=None):
    gpu = CvoGPU(params=P)
    ell = P.ell_init if ell is None else ell
    K = P.nearest_neighbors_max if K is None else K
    g = gpu.align(src, tgt, init, max_iterations=1, ell0=ell, K0=K, trace_capacity=2, trace_dense=2)
    mat, ind, nz = gpu.debug_last_ell(src.num_points(), K)
    o = oracle.iteration(oracle.params_from(P), _ocloud(oracle, src), _ocloud(oracle, tgt), init[:3, :3],
                         init[:3, 3], ell, K, want_ell=True)
    assert np.array_equal(nz, o["nonzeros"])          # bit-exact: index work
    assert np.array_equal(ind, o["ind"])              # bit-exact: ordered truncation included
    assert np.allclose(mat, o["mat"], rtol=2e-7, atol=0)  # 1 float ulp: exp() of two libms
    assert len(g.trace) == 1
    _cmp_trace(g.trace[0], o["trace"])
    assert gpu.debug_last_candidates() >= int(nz.sum())  # the scan is a superset of the association
    return g, o, (mat, ind, nz)


@pytest.mark.parametrize("n,m", [(1000, 1000), (777, 1234), (64, 5000), (3000, 65)])
def test_single_iteration_geometric(oracle, n, m):
    P, src, tgt, init = cases.config2(n=n, m=m)
    _single_iteration(oracle, P, src, tgt, init)


def test_single_iteration_nonidentity_state(oracle):
    P, src, tgt, _ = cases.config2(n=1500)
    init = synth.gt_motion().astype(np.float32)
    _single_iteration(oracle, P, src, tgt, init, ell=0.12, K=9)


def test_single_iteration_ordered_truncation(oracle):
    """Rows over the cap keep the first K qualifying targets in ascending j."""
    P, src, tgt, init = cases.config2(n=600)
    g, o, (mat, ind, nz) = _single_iteration(oracle, P, src, tgt, init, ell=1.5, K=5)
    assert (nz == 5).all() and np.all(np.diff(ind, axis=1) > 0)


def test_single_iteration_colour(oracle):
    P, src, tgt, init = cases.config3(n=1500)
    _single_iteration(oracle, P, src, tgt, init)


def test_single_iteration_semantic(oracle):
    P, src, tgt, init = cases.config4(n=1500)
    _single_iteration(oracle, P, src, tgt, init)
    _single_iteration(oracle, P, src, tgt, init, ell=0.6, K=40)


def test_single_iteration_geometric_type_and_range_ell(oracle):
    P, src, tgt, init = cases.config2(n=900)
    P.is_using_geometric_type = 1
    P.is_using_range_ell = 1
    rs = np.random.default_rng(5)
    gx = np.where(rs.random((900, 1)) < 0.5, [[1.0, 0.0]], [[0.0, 1.0]]).astype(np.float32)
    gy = np.where(rs.random((900, 1)) < 0.5, [[1.0, 0.0]], [[0.3, 0.9]]).astype(np.float32)
    src = CvoPointCloud.from_arrays(src.positions(), None, None, gx)
    tgt = CvoPointCloud.from_arrays(tgt.positions(), None, None, gy)
    _single_iteration(oracle, P, src, tgt, init, ell=0.5)


def test_no_geometry_kernel_is_dense(oracle):
    """is_using_geometry = 0: k = 1 and no distance cut at all (every pair is a candidate)."""
    P, src, tgt, init = cases.config3(n=150)
    P.is_using_geometry = 0
    _single_iteration(oracle, P, src, tgt, init, K=64)


def test_full_size_single_iteration_10k(oracle):
    """BASELINE.json's headline size, checked directly (the oracle needs milliseconds for one pass)."""
    P, src, tgt, init = cases.config2(n=10000)
    _single_iteration(oracle, P, src, tgt, init)


def _prefix(oracle, P, src, tgt, init, n_it, gpu=None):
    gpu = gpu or CvoGPU(params=P)
    g = gpu.align(src, tgt, init, max_iterations=n_it, trace_capacity=n_it, trace_dense=n_it)
    o = oracle.align(oracle.params_from(P), _ocloud(oracle, src), _ocloud(oracle, tgt), init,
                     trace_capacity=n_it, trace_dense=n_it, max_iterations=n_it)
    assert g.iterations == o["iterations"] and g.ret == o["ret"]
    assert len(g.trace) == len(o["trace"])
    return g, o


@pytest.mark.parametrize("builder,kw,n_it", [(cases.config2, dict(n=2000), 120), (cases.config3, dict(n=1500), 80),
                                             (cases.config4, dict(n=2000), 80), (cases.config1, {}, 150),
                                             (cases.scene, dict(n=3000), 150)])
def test_trajectory_prefix(oracle, builder, kw, n_it):
    """Per-iteration state over a prefix of the optimisation (config 1 runs on the neighbour cap throughout)."""
    P, src, tgt, init = builder(**kw)
    g, o = _prefix(oracle, P, src, tgt, init, n_it)
    for a, b in zip(g.trace, o["trace"]):
        _cmp_trace(a, b)
    assert cases.max_abs_diff(g.transform, o["transform"]) <= 1e-6


@pytest.mark.parametrize("builder,n_it", [(cases.config3, 220), (cases.config4, 220), (cases.config2, 220), (cases.scene, 100)])
def test_trajectory_prefix_full_size_10k(oracle, builder, n_it):
    """BASELINE.json configs 3 / 4 (and the 10k geometric shape of config 5) at their FULL size: every iteration of a
    220-iteration prefix - the fast first iterations with their list rebuilds, then the lean graph - must take the
    oracle's integer decisions (K, nnz, max_nnz, ell) and follow its twist / coefficients / pose."""
    P, src, tgt, init = builder(n=10000)
    g, o = _prefix(oracle, P, src, tgt, init, n_it)
    for a, b in zip(g.trace, o["trace"]):
        _cmp_trace(a, b)
    assert cases.max_abs_diff(g.transform, o["transform"]) <= 1e-6


def test_final_pose_config4_full_size(oracle):
    """Config 4 at 10k x 10k to its own dist < eps_2 stop (~300 iterations): same stop iteration, pose to 1e-4."""
    P, src, tgt, init = cases.config4(n=10000)
    g = CvoGPU(params=P).align(src, tgt, init)
    o = oracle.align(oracle.params_from(P), _ocloud(oracle, src), _ocloud(oracle, tgt), init)
    assert g.ret == o["ret"] == 0 and g.iterations < P.MAX_ITER
    assert abs(g.iterations - o["iterations"]) <= 2
    assert cases.max_abs_diff(g.transform, o["transform"]) <= TOL_POSE


def test_final_pose_config2_clamped(oracle):
    P, src, tgt, init = cases.config2(n=2000)
    gpu = CvoGPU(params=P)
    g = gpu.align(src, tgt, init)
    o = oracle.align(oracle.params_from(P), _ocloud(oracle, src), _ocloud(oracle, tgt), init)
    assert g.iterations == o["iterations"] == P.MAX_ITER and g.ret == o["ret"] == 0
    assert cases.max_abs_diff(g.transform, o["transform"]) <= TOL_POSE_CLAMPED
    assert cases.max_abs_diff(g.transform, np.linalg.inv(synth.gt_motion())) < 2e-3
    assert g.seconds > 0


def test_final_pose_config4_converges(oracle):
    """Warm-start semantic tracking ends through dist < eps_2: well conditioned, 1e-4."""
    P, src, tgt, init = cases.config4(n=2000)
    gpu = CvoGPU(params=P)
    g = gpu.align(src, tgt, init)
    o = oracle.align(oracle.params_from(P), _ocloud(oracle, src), _ocloud(oracle, tgt), init)
    assert g.ret == o["ret"] == 0
    assert abs(g.iterations - o["iterations"]) <= 2 and g.iterations < P.MAX_ITER
    assert cases.max_abs_diff(g.transform, o["transform"]) <= TOL_POSE


def test_config1_demo_fixed_iteration_pose(oracle):
    """Config 1's own stop is an accidental small step: compare at k = 1000."""
    P, src, tgt, init = cases.config1()
    g, o = _prefix(oracle, P, src, tgt, init, 1000, gpu=CvoGPU(params=P))
    assert g.trace[0].max_nnz == 256 == g.trace[0].K
    assert cases.max_abs_diff(g.transform, o["transform"]) <= TOL_POSE


def test_against_committed_golden_traces():
    """HIP path vs tests/golden/oracle_traces.json (no oracle call at all)."""
    with open(os.path.join(cases.GOLDEN, "oracle_traces.json")) as f:
        gold = {c["name"]: c for c in json.load(f)["cases"]}
    for name, builder in (("config2_n2000", cases.config2), ("config4_n2000", cases.config4)):
        gc = gold[name]
        P, src, tgt, init = builder(**gc["kwargs"])
        gpu = CvoGPU(params=P)
        g = gpu.align(src, tgt, init, trace_capacity=400, trace_dense=50, trace_every=100)
        got = {t.k: t for t in g.trace}
        for t in gc["trace"]:
            if t["k"] >= 50:
                continue  # beyond the dense prefix trajectories may differ in the last bits
            a = got[t["k"]]
            assert (a.K, a.nnz, a.max_nnz) == (t["K"], t["nnz"], t["max_nnz"]), (name, t["k"])
            assert np.allclose(list(a.omega) + list(a.v), t["omega"] + t["v"], atol=TOL_TWIST)
        tol = TOL_POSE if name == "config4_n2000" else TOL_POSE_CLAMPED
        assert cases.max_abs_diff(g.transform, gc["transform"]) <= tol
        ip = gpu.inner_product_gpu(src, tgt, init, P.ell_init)
        assert ip == pytest.approx(gc["inner_product_init"], rel=TOL_IP_REL)
        fa = gpu.function_angle(src, tgt, init, P.ell_init, True)
        assert fa == pytest.approx(gc["function_angle_init"], rel=TOL_IP_REL)


@pytest.mark.parametrize("builder,kw", [(cases.config2, dict(n=3000)), (cases.config4, dict(n=1500)),
                                        (cases.config1, {})])
def test_inner_product_and_function_angle(oracle, builder, kw):
    P, src, tgt, init = builder(**kw)
    gpu = CvoGPU(params=P)
    op = oracle.params_from(P)
    ox, oy = _ocloud(oracle, src), _ocloud(oracle, tgt)
    for T, ell in ((init, P.ell_init), (synth.gt_motion().astype(np.float32), 0.25)):
        ip_g = gpu.inner_product_gpu(src, tgt, T, ell)
        ip_o = oracle.inner_product(op, ox, oy, T, ell)
        assert ip_g == pytest.approx(ip_o, rel=TOL_IP_REL, abs=1e-12)
        for approx in (True, False):
            fa_g = gpu.function_angle(src, tgt, T, ell, approx)
            fa_o = oracle.function_angle(op, ox, oy, T, ell, approx)
            assert fa_g == pytest.approx(fa_o, rel=TOL_IP_REL, abs=1e-12)


def test_function_angle_self_is_one():
    P, src, _, _ = cases.config2(n=4000)
    gpu = CvoGPU(params=P)
    assert gpu.function_angle(src, src, np.eye(4), 0.2, False) == pytest.approx(1.0, abs=1e-6)


def test_inner_product_is_additive_over_target_subsets():
    """Linearity property at BASELINE size: <X, Y1 u Y2> = <X, Y1> + <X, Y2> while no row hits K_max."""
    P, src, tgt, init = cases.config2(n=10000)
    gpu = CvoGPU(params=P)
    y = tgt.positions()
    a = gpu.inner_product_gpu(src, CvoPointCloud.from_xyz(y[:6000]), init, 0.3)
    b = gpu.inner_product_gpu(src, CvoPointCloud.from_xyz(y[6000:]), init, 0.3)
    c = gpu.inner_product_gpu(src, tgt, init, 0.3)
    assert c == pytest.approx(a + b, rel=1e-5)


def test_association_matches_oracle(oracle):
    P, src, tgt, init = cases.config2(n=1200)
    gpu = CvoGPU(params=P)
    rp, col, val = gpu.compute_association_gpu(src, tgt, init, 0.3)
    orp, ocol, oval = oracle.association(oracle.params_from(P), _ocloud(oracle, src), _ocloud(oracle, tgt), init, 0.3)
    assert np.array_equal(rp, orp) and np.array_equal(col, ocol)
    assert np.allclose(val, oval, rtol=2e-7, atol=0)


def test_empty_cloud_returns_zero_and_leaves_transform():
    gpu = CvoGPU(params=CvoParams())
    empty = CvoPointCloud.from_xyz(np.zeros((0, 3), np.float32))
    full = CvoPointCloud.from_xyz(np.ones((10, 3), np.float32))
    r = gpu.align(empty, full, np.eye(4))
    assert r.ret == 0 and r.transform is None
    assert gpu.function_angle(gpu.upload(empty), gpu.upload(full), np.eye(4), 0.3) == 0.0


def test_flow_vanishes_returns_minus_one(oracle):
    """Clouds farther apart than the cut-off: empty A, ret = -1, pose untouched (CvoGPU.cu:1454-1458)."""
    P, src, tgt, init = cases.config2(n=500)
    far = CvoPointCloud.from_xyz(tgt.positions() + np.float32(100.0))
    gpu = CvoGPU(params=P)
    g = gpu.align(src, far, init)
    assert g.ret == -1 and g.iterations == 0
    assert np.allclose(g.transform, np.eye(4))
    # NaN geometric types: every comparison false
    P.is_using_geometric_type = 1
    g = CvoGPU(params=P).align(CvoPointCloud.from_arrays(src.positions()), CvoPointCloud.from_arrays(tgt.positions()), init)
    assert g.ret == -1


def test_tiny_clouds(oracle):
    P, _, _, init = cases.config2(n=100)
    a = CvoPointCloud.from_xyz(np.array([[0.0, 0.0, 5.0]], np.float32))
    b = CvoPointCloud.from_xyz(np.array([[0.05, 0.0, 5.0], [9.0, 9.0, 9.0]], np.float32))
    gpu = CvoGPU(params=P)
    g = gpu.align(a, b, init, max_iterations=20, trace_capacity=20, trace_dense=20)
    o = oracle.align(oracle.params_from(P), _ocloud(oracle, a), _ocloud(oracle, b), init, trace_capacity=20,
                     trace_dense=20, max_iterations=20)
    assert g.iterations == o["iterations"] and g.ret == o["ret"]
    assert cases.max_abs_diff(g.transform, o["transform"]) <= 1e-6


def test_unsupported_and_invalid_arguments():
    P, src, tgt, init = cases.config2(n=100)
    P.is_using_kdtree = 1
    with pytest.raises(CvoError):
        CvoGPU(params=P).align(src, tgt, init)
    P.is_using_kdtree = 0
    P.nearest_neighbors_max = 0
    with pytest.raises(CvoError):
        CvoGPU(params=P).align(src, tgt, init)


def test_graph_and_plain_launch_paths_agree():
    P, src, tgt, init = cases.config2(n=1500)
    gpu = CvoGPU(params=P)
    a = gpu.align(src, tgt, init, max_iterations=333, use_graph=1)
    b = gpu.align(src, tgt, init, max_iterations=333, use_graph=2, iters_per_launch=7)
    assert a.iterations == b.iterations == 333
    assert np.array_equal(a.transform, b.transform)


def test_batch_equals_individual_aligns():
    """cvo_align_batch: ragged pairs solved concurrently give bit-identical poses to one-at-a-time calls."""
    pairs = [cases.config2(n=1500, pair_id=0), cases.config2(n=900, pair_id=1, m=1300), cases.config2(n=2000, pair_id=2)]
    P = pairs[0][0]
    gpu = CvoGPU(params=P)
    res = gpu.align_batch([p[1] for p in pairs], [p[2] for p in pairs], [p[3] for p in pairs], max_iterations=400)
    for (Pp, s, t, init), r in zip(pairs, res):
        one = CvoGPU(params=P).align(s, t, init, max_iterations=400)
        assert r.iterations == one.iterations == 400
        assert np.array_equal(r.transform, one.transform)


def _ragged(builder, k, n_lo, n_hi, seed):
    rs = np.random.default_rng(seed)
    out = []
    for p in range(k):
        n = int(rs.integers(n_lo, n_hi))
        P, a, b, init = builder(n=n, pair_id=p)
        if p % 3 == 1:  # ragged: fewer source rows than targets, and the other way round
            xs, fs, ls, gs = a.device_arrays()
            cut = int(n * 0.7)
            a = CvoPointCloud.from_arrays(xs[:cut], None if fs is None else fs[:cut], None if ls is None else ls[:cut], gs[:cut])
        elif p % 3 == 2:
            xt, ft, lt, gt = b.device_arrays()
            cut = int(n * 0.8)
            b = CvoPointCloud.from_arrays(xt[:cut], None if ft is None else ft[:cut], None if lt is None else lt[:cut], gt[:cut])
        out.append((P, a, b, init))
    return out


@pytest.mark.parametrize("builder,n_it", [(cases.config3, 260), (cases.config4, 10000)])
def test_batch_of_32_general_pairs_equals_individual_aligns(builder, n_it):
    """The GENERAL (colour / semantic) instantiation under four concurrent sub-batch streams and the lean graph: 32
    ragged pairs solved as one batch give bit-identical poses, iteration counts and final ell / K to 32 separate
    cvo_align calls (config 4 runs to its own eps_2 stops, which differ from pair to pair)."""
    pairs = _ragged(builder, 32, 1200, 3200, seed=42)
    P = pairs[0][0]
    gpu = CvoGPU(params=P)
    res = gpu.align_batch([p[1] for p in pairs], [p[2] for p in pairs], [p[3] for p in pairs], max_iterations=n_it)
    n_groups, _ = gpu.debug_last_geometry()
    assert n_groups == 4
    solo = CvoGPU(params=P)
    its = set()
    for (Pp, s_, t_, init), r in zip(pairs, res):
        one = solo.align(s_, t_, init, max_iterations=n_it)
        assert (r.iterations, r.ret, r.final_ell, r.final_num_neighbors) == (one.iterations, one.ret, one.final_ell,
                                                                             one.final_num_neighbors)
        assert np.array_equal(r.transform, one.transform)
        its.add(r.iterations)
    if builder is cases.config4:
        assert len(its) > 4 and max(its) < 10000     # really ended by eps_2, at different iterations


def test_determinism_full_size():
    P, src, tgt, init = cases.config2(n=10000)
    gpu = CvoGPU(params=P)
    a = gpu.align(src, tgt, init, max_iterations=150)
    b = gpu.align(src, tgt, init, max_iterations=150)
    assert np.array_equal(a.transform, b.transform)


def test_noise_free_alignment_recovers_motion_full_size():
    """Size-independent property at 10k x 10k: with exact correspondences the returned transform is T_gt^-1."""
    P = cases.load_params("geometric_gpu")
    src, tgt, _ = synth.geometric_pair(10000, 7, noise=0.0)
    gpu = CvoGPU(params=P)
    g = gpu.align(CvoPointCloud.from_xyz(src), CvoPointCloud.from_xyz(tgt), np.eye(4))
    assert g.ret == 0
    # the loop ends jittering by +-min_step along a unit twist: algorithmic, not parity
    assert cases.max_abs_diff(g.transform, np.linalg.inv(synth.gt_motion())) < 3e-3


def test_wide_target_cloud_uses_32bit_candidate_lists(oracle):
    """M >= 65536 switches k_assoc to its 32-bit index instantiation; ragged N << M."""
    rs = np.random.default_rng(11)
    P = cases.load_params("geometric_gpu")
    m = 70000
    tgt = np.stack([rs.uniform(-40, 40, m), rs.uniform(-2, 2, m), rs.uniform(2, 82, m)], axis=1).astype(np.float32)
    src = (tgt[rs.choice(m, 300, replace=False)] + rs.normal(0, 0.02, (300, 3))).astype(np.float32)
    _single_iteration(oracle, P, CvoPointCloud.from_xyz(src), CvoPointCloud.from_xyz(tgt), np.eye(4, dtype=np.float32))


def test_candidate_list_overflow_takes_literal_path(oracle):
    """More candidates per row than the LDS list holds, but fewer hits than K: the literal ordered scan must
    produce the same ELL as the sorted-list path would (mid-density regime)."""
    P, src, tgt, init = cases.config2(n=3000)
    g, o, (mat, ind, nz) = _single_iteration(oracle, P, src, tgt, init, ell=1.0, K=512)
    assert nz.max() > 64  # beyond ASSOC_CAP16


def test_result_does_not_depend_on_spatial_order(monkeypatch):
    """The k-d ordering only steers the tile culling: identity order gives bit-identical poses."""
    P, src, tgt, init = cases.config2(n=2500)
    a = CvoGPU(params=P).align(src, tgt, init, max_iterations=200)
    monkeypatch.setenv("CVO_NO_SORT", "1")
    b = CvoGPU(params=P).align(src, tgt, init, max_iterations=200)
    assert np.array_equal(a.transform, b.transform)


def test_result_does_not_depend_on_list_reuse(monkeypatch):
    """Candidate lists are reused across iterations (scan with a skin, lean graph): a superset of the exact test,
    so scanning every iteration (CVO_SKIN=0) or never using the lean graph gives bit-identical poses."""
    P, src, tgt, init = cases.config2(n=3000)
    gpu = CvoGPU(params=P)
    a = gpu.align(src, tgt, init, max_iterations=600)
    builds, iters, cand = gpu.debug_list_builds()
    assert iters == 600 and 0 < builds < 300      # the lists really were reused
    monkeypatch.setenv("CVO_NO_LEAN", "1")
    b = CvoGPU(params=P).align(src, tgt, init, max_iterations=600)
    monkeypatch.setenv("CVO_SKIN", "0")
    g0 = CvoGPU(params=P)
    c = g0.align(src, tgt, init, max_iterations=600)
    assert g0.debug_list_builds()[0] >= 600       # one scan per iteration (+ the request after the last one)
    assert a.iterations == b.iterations == c.iterations == 600
    assert np.array_equal(a.transform, b.transform)
    assert np.array_equal(a.transform, c.transform)


def test_batch_scheduling_switches_do_not_change_results(monkeypatch):
    """In a batch the optional (ell-shrink) rebuilds wait for common iteration counts, lists that would not reach the
    next of them are renewed early, chunks grow from 16 to 32 iterations after the first 256 and the two blind first
    chunks of a call are 4 iterations long: scheduling only - the poses are bit-identical with all of it switched off."""
    cs = [cases.config2(n=2000, pair_id=p) for p in range(8)]
    P = cs[0][0]
    runs = []
    for env in ({}, {"CVO_SHRINK_ALIGN": "0"}, {"CVO_FIXED_CHUNKS": "1", "CVO_SHRINK_ALIGN": "15"},
                {"CVO_FIRST_U": "16"}, {"CVO_FIRST_U": "2", "CVO_FIRST_CHUNKS": "6"}):   # the blind first chunks of a call
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        gpu = CvoGPU(params=P)
        res = gpu.align_batch([c[1] for c in cs], [c[2] for c in cs], [c[3] for c in cs], max_iterations=700)
        runs.append((res, gpu.debug_list_builds()[0]))
        for k in env:
            monkeypatch.delenv(k)
    for res, _ in runs[1:]:
        for a, b in zip(runs[0][0], res):
            assert a.iterations == b.iterations == 700 and np.array_equal(a.transform, b.transform)
    assert len({b for _, b in runs}) > 1          # the schedules really differed (list builds)


def test_lean_graph_waits_do_not_change_results(monkeypatch):
    """A rebuild opportunity only every 16 iterations makes pairs wait for their next list; the trajectory is the
    same as with an opportunity in every iteration."""
    P, src, tgt, init = cases.config2(n=2000)
    a = CvoGPU(params=P).align(src, tgt, init, max_iterations=500)
    monkeypatch.setenv("CVO_LEAN_U", "16")
    b = CvoGPU(params=P).align(src, tgt, init, max_iterations=500)
    monkeypatch.setenv("CVO_LEAN_U", "3")
    c = CvoGPU(params=P).align(src, tgt, init, max_iterations=500)
    assert a.iterations == b.iterations == c.iterations == 500
    assert np.array_equal(a.transform, b.transform)
    assert np.array_equal(a.transform, c.transform)


def test_one_hot_semantics_equal_the_general_kernel_bit_for_bit():
    """Config 4's class rows are exact one-hot rows: the upload finds that out and the association kernels compare 4-byte
    class ids (squared class distance 0 or exactly 2, the semantic kernel one of two constants: FEAT_HOT) instead of
    streaming two 80-byte rows per surviving pair (CvoGPU.cu:563-569).  Same bits as the general kernel (NO_ONEHOT), in
    the loop, in the inner product and in the exported matrix; a soft row anywhere sends the call down the general path."""
    P, a, b, init = cases.config4(n=4000)
    gpu = CvoGPU(params=P)
    da, db = gpu.upload(a), gpu.upload(b)
    hot = gpu.align(da, db, init, max_iterations=120)
    ip_hot = gpu.inner_product_gpu(da, db, init, 0.25)
    A_hot = gpu.compute_association_gpu(da, db, init, 0.25)
    gpu.set_option("NO_ONEHOT", "1")
    gen = gpu.align(da, db, init, max_iterations=120)
    assert hot.iterations == gen.iterations == 120 and np.array_equal(hot.transform, gen.transform)
    assert (hot.final_ell, hot.final_num_neighbors) == (gen.final_ell, gen.final_num_neighbors)
    assert ip_hot == gpu.inner_product_gpu(da, db, init, 0.25)
    A_gen = gpu.compute_association_gpu(da, db, init, 0.25)
    assert all(np.array_equal(x, y) for x, y in zip(A_hot, A_gen)) and len(A_hot[1]) > 1000
    gpu.set_option("NO_ONEHOT", None)
    # one soft row in the target: no class ids for that cloud, the general kernel runs by itself
    lab = np.array(b.labels(), np.float32).copy()
    lab[7] = 0.0
    lab[7, :2] = 0.5
    b2 = type(b).from_arrays(np.array(b.positions()), np.array(b.features()), lab, np.array(b.geometric_types()).reshape(-1, 2))
    soft = gpu.align(da, gpu.upload(b2), init, max_iterations=120)
    assert soft.iterations == 120 and not np.array_equal(soft.transform, hot.transform)
    assert cases.max_abs_diff(soft.transform, hot.transform) < 1e-3


def test_overlap_kernel_equals_the_list_chain():
    """inner_product_gpu / function_angle in one launch (k_overlap: 64-row blocks against the target tiles their bounding
    spheres reach, the reference's per-pair arithmetic on what passes the cut-off) against the list chain the loop uses
    (CVO_IP_CHAIN): the same values summed in another order - equal to a few ulp of the double sum, i.e. the same float
    in all but exceptional cases.  Geometry only, colour, soft and one-hot semantics, ragged sizes, a tiny cloud, poses
    away from the identity, a matrix that is no rotation; and rows beyond nearest_neighbors_max, where the first-K
    truncation (CvoGPU.cu:585) sends the call down the chain by itself: then bit for bit."""
    rng = np.random.default_rng(5)

    def pose(angle, t):
        c, s_ = np.cos(angle), np.sin(angle)
        T = np.eye(4, dtype=np.float32)
        T[:3, :3] = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]], np.float32)
        T[:3, 3] = t
        return T

    work = [("geo 3000", cases.config2(n=3000), 0.3), ("geo ragged", cases.config2(n=1000, m=777), 0.25),
            ("colour 2000", cases.config3(n=2000), None), ("semantic 2000", cases.config4(n=2000), None),
            ("geo 10k", cases.config2(n=10000), None), ("geo tiny", cases.config2(n=5, m=3), 0.8)]
    for name, (P, src, tgt, init), ell in work:
        ell = P.ell_init if ell is None else ell
        gpu = CvoGPU(params=P)
        da, db = gpu.upload(src), gpu.upload(tgt)
        skew = np.eye(4, dtype=np.float32)
        skew[0, 0], skew[1, 2] = 1.3, 0.2  # (not a rotation: the cull may only trust its norm)
        for T in (init, pose(0.05, (0.02, -0.01, 0.03)), pose(-0.4, (0.3, 0.1, -0.2)), skew):
            fast = (gpu.inner_product_gpu(da, db, T, ell), gpu.function_angle(da, db, T, ell, True), gpu.function_angle(da, db, T, ell, False))
            gpu.set_option("IP_CHAIN", "1")
            chain = (gpu.inner_product_gpu(da, db, T, ell), gpu.function_angle(da, db, T, ell, True), gpu.function_angle(da, db, T, ell, False))
            gpu.set_option("IP_CHAIN", None)
            assert fast == pytest.approx(chain, rel=2e-7, abs=1e-30), (name, fast, chain)
            assert fast == (gpu.inner_product_gpu(da, db, T, ell), gpu.function_angle(da, db, T, ell, True), gpu.function_angle(da, db, T, ell, False))
        if name == "semantic 2000":  # class ids against class rows
            hot = gpu.inner_product_gpu(da, db, init, ell)
            gpu.set_option("NO_ONEHOT", "1")
            assert hot == gpu.inner_product_gpu(da, db, init, ell)
        gpu.close()
    # far from the origin, under a pose whose translation cancels the rotated coordinates (the cull's rounding slack is
    # measured on the terms of Rinv y + Tinv, not on the result)
    P, src, tgt, init = cases.config2(n=3000)
    off = np.array([800.0, -300.0, 150.0], np.float32)
    xs, ys = np.array(src.positions()) + off, np.array(tgt.positions()) + off
    R = pose(0.05, (0, 0, 0))[:3, :3].astype(np.float64)
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = R
    T[:3, 3] = (off.astype(np.float64) - R @ off.astype(np.float64)).astype(np.float32)
    gpu = CvoGPU(params=P)
    da, db = gpu.upload(CvoPointCloud.from_xyz(xs)), gpu.upload(CvoPointCloud.from_xyz(ys))
    fast = gpu.inner_product_gpu(da, db, T, 0.3)
    gpu.set_option("IP_CHAIN", "1")
    assert fast == pytest.approx(gpu.inner_product_gpu(da, db, T, 0.3), rel=2e-7) and fast > 0
    gpu.close()
    # rows that find more than K pairs: the sum is the chain's, bit for bit
    P, src, tgt, init = cases.config2(n=2000)
    P.nearest_neighbors_max = 6
    gpu = CvoGPU(params=P)
    da, db = gpu.upload(src), gpu.upload(tgt)
    fast = gpu.inner_product_gpu(da, db, init, 0.5)
    gpu.set_option("IP_CHAIN", "1")
    assert fast == gpu.inner_product_gpu(da, db, init, 0.5) and fast > 0
    A = gpu.compute_association_gpu(da, db, init, 0.5)
    assert np.diff(A[0]).max() == 6  # (rows sit on the cap)


def test_association_export_reads_row_major_runs(oracle):
    """Rows beyond their lists keep their ELL entries row-major (PairDesc::dense_off, laid out by k_list); the exports have to
    find them there.  A clustered scene, rows of several hundred entries: compute_association_gpu against the oracle's
    matrix (columns exact, values to an ulp), its sum against inner_product_gpu through the list chain, and the last
    iteration's matrix of an align() against its own nonzero count."""
    P, src, tgt, init = cases.scene(n=2500)
    gpu = CvoGPU(params=P)
    da, db = gpu.upload(src), gpu.upload(tgt)
    ell = 0.8
    rp, col, val = gpu.compute_association_gpu(da, db, init, ell)[:3]
    counts = np.diff(rp)
    assert counts.max() > 200 and (counts > 64).sum() > 100          # (rows that live in the row-major part)
    orp, ocol, oval = oracle.association(oracle.params_from(P), _ocloud(oracle, src), _ocloud(oracle, tgt), init, ell)
    assert np.array_equal(rp, orp) and np.array_equal(col, ocol)
    assert np.allclose(val, oval, rtol=2e-7, atol=0)
    gpu.set_option("IP_CHAIN", "1")
    assert float(np.sum(val.astype(np.float64))) == pytest.approx(gpu.inner_product_gpu(da, db, init, ell), rel=1e-6)
    gpu.set_option("IP_CHAIN", None)
    g = gpu.align(da, db, init, max_iterations=30, trace_capacity=30, trace_dense=30)
    rp2, col2, val2 = gpu.align_association(src.num_points())[:3]
    assert len(col2) == g.trace[-1].nnz and np.diff(rp2).max() > 64 and np.all(val2 > 0)


def test_context_options_are_validated():
    gpu = CvoGPU(params=CvoParams())
    gpu.set_option("CVO_VERBOSE", None)      # with or without the prefix; None clears
    gpu.set_option("SKIN", "1.5")
    with pytest.raises(CvoError, match="unknown option"):
        gpu.set_option("NO_SUCH_SWITCH", "1")


def test_timing_replay_leaves_the_context_usable():
    """cvo_debug_time_kernels / cvo_debug_time_scan replay launches without writing anything back."""
    P, src, tgt, init = cases.config2(n=2000)
    gpu = CvoGPU(params=P)
    a = gpu.align(src, tgt, init, max_iterations=200)
    t_assoc, t_coeff = gpu.debug_time_kernels(3)
    t_scan = gpu.debug_time_scan(3)
    assert t_assoc > 0 and t_coeff > 0 and t_scan > 0
    b = gpu.align(src, tgt, init, max_iterations=200)
    assert np.array_equal(a.transform, b.transform)
    ip0 = gpu.inner_product_gpu(src, tgt, init, P.ell_init)
    assert ip0 == gpu.inner_product_gpu(src, tgt, init, P.ell_init)


def test_kernel_clock_and_phase_stamps_do_not_change_results(monkeypatch):
    """CVO_KERNEL_CLOCK / CVO_PHASE_TICKS only observe: same trajectory, and the in-loop clock reports plausible
    per-launch durations for both per-iteration kernels."""
    P, src, tgt, init = cases.config2(n=3000)
    ref = CvoGPU(params=P).align(src, tgt, init, max_iterations=300)
    monkeypatch.setenv("CVO_KERNEL_CLOCK", "1")
    monkeypatch.setenv("CVO_PHASE_TICKS", "1")
    gpu = CvoGPU(params=P)
    a = gpu.align(src, tgt, init, max_iterations=300)
    assert a.iterations == ref.iterations and np.array_equal(a.transform, ref.transform)
    t_assoc, t_coeff, n = gpu.debug_kernel_clock()
    assert 250 <= n <= 300                       # one interval per iteration (a few may run in the full graph)
    assert 5e-4 < t_assoc < 1.0 and 5e-4 < t_coeff < 1.0   # ms: between half a microsecond and a millisecond
    gpu.debug_time_kernels(2)                    # prints the phase stamps to stderr
    monkeypatch.delenv("CVO_KERNEL_CLOCK")
    gpu2 = CvoGPU(params=P)
    gpu2.align(src, tgt, init, max_iterations=50)
    with pytest.raises(Exception):
        gpu2.debug_kernel_clock()


def test_kernel_clock_per_call_option_alternates_with_production_kernels():
    """cvo_align_opts_t.kernel_clock times single calls of a loop (bench.py: the last timed step): instrumented and
    production calls alternate on one context without disturbing each other (both sets of graphs stay cached) and
    return the same bits."""
    P, src, tgt, init = cases.config2(n=3000)
    gpu = CvoGPU(params=P)
    s, t = gpu.upload_many([src, tgt])
    ref = gpu.align(s, t, init, max_iterations=200)
    for clocked in (True, False, True, False):
        a = gpu.align(s, t, init, max_iterations=200, kernel_clock=clocked)
        assert a.iterations == ref.iterations and np.array_equal(a.transform, ref.transform)
        if clocked:
            t_assoc, t_coeff, n = gpu.debug_kernel_clock()
            assert 150 <= n <= 200 and 5e-4 < t_assoc < 1.0 and 5e-4 < t_coeff < 1.0
        else:
            with pytest.raises(Exception):
                gpu.debug_kernel_clock()


@pytest.mark.parametrize("env", [{}, {"CVO_NO_DENSE_REGIME": "1"}, {"CVO_NO_LEAN": "1"}])
def test_config1_is_reproducible_run_to_run(monkeypatch, env):
    """The demo pair stops on an accidentally small step: one flipped bit anywhere changes its iteration
    count, so repeated runs expose any cross-block race of the last-block patterns (flow gate, update) - the version
    without a vmcnt wait in front of the counters failed one run in three on the list + overflow path."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    P, src, tgt, init = cases.config1()
    gpu = CvoGPU(params=P)
    runs = [gpu.align(src, tgt, init) for _ in range(6)]
    assert len({r.iterations for r in runs}) == 1
    assert all(np.array_equal(r.transform, runs[0].transform) for r in runs)
    assert runs[0].iterations == 6661


def test_contexts_and_clouds_release_device_memory():
    """Twenty context / cloud / align cycles leave the free device memory where it was (a 64-pair workspace is
    an 8-pair workspace alone is ~100 MB
    per cycle: a leak would show at once; scripts/leak_check.py prints the curve)."""
    import gc
    P, src, tgt, init = cases.config2(n=4000)
    probe = CvoGPU(params=P)   # (hipMemGetInfo through the C-ABI: no second HIP runtime - torch's - in this process)

    def cycle():
        gpu = CvoGPU(params=P)
        ds, dt = gpu.upload(src), gpu.upload(tgt)
        r = gpu.align_batch([ds] * 8, [dt] * 8, [init] * 8, max_iterations=40)
        ds.free()
        dt.free()
        gpu.close() if hasattr(gpu, "close") else None
        del gpu, ds, dt
        gc.collect()
        return r[0].transform

    first = cycle()
    for _ in range(4):  # (the HIP runtime grows its own pools in 64 MiB steps during the first cycles)
        cycle()
    free0 = probe.debug_device_memory()[0]
    for _ in range(20):
        assert np.array_equal(cycle(), first)
    free1 = probe.debug_device_memory()[0]
    # (one cycle allocates ~0.4 GB; the runtime's own pools move in steps of a few hundred MiB at most)
    assert free0 - free1 < 1 << 30, f"device memory shrank by {(free0 - free1) >> 20} MiB over 20 cycles"


def test_upload_many_matches_single_uploads():
    P, src, tgt, init = cases.config2(n=1500)
    gpu = CvoGPU(params=P)
    a = gpu.align(src, tgt, init, max_iterations=60)
    ds = gpu.upload_many([src, tgt, src, tgt], threads=4)
    b = gpu.align(ds[0], ds[1], init, max_iterations=60)
    c = gpu.align(ds[2], ds[3], init, max_iterations=60)
    assert np.array_equal(a.transform, b.transform) and np.array_equal(a.transform, c.transform)


def test_dense_regime_switches_both_ways(oracle):
    """Config 1 starts with nearly every row on K_max (all rows served by k_assoc_dense, lists never rebuilt) and
    thins out later: the list builds stay a handful, and the trajectory is the oracle's (test_config1_* cover the
    bits; here only the mechanism is observed)."""
    P, src, tgt, init = cases.config1()
    gpu = CvoGPU(params=P)
    g = gpu.align(src, tgt, init, max_iterations=1000)
    builds, iters, _ = gpu.debug_list_builds()
    assert iters >= 1000 and builds <= 8
    o = oracle.align(oracle.params_from(P), _ocloud(oracle, src), _ocloud(oracle, tgt), init, max_iterations=1000)
    assert g.iterations == o["iterations"]
    assert cases.max_abs_diff(g.transform, o["transform"]) == 0.0


def _pose34(angle_deg, axis, t):
    a = np.asarray(axis, np.float64) / np.linalg.norm(axis)
    th = np.deg2rad(angle_deg)
    Kx = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    return np.hstack([R, np.asarray(t, np.float64).reshape(3, 1)])


@pytest.mark.parametrize("builder,kw,K", [(cases.config2, dict(n=1500), 64), (cases.config3, dict(n=1200), 32),
                                          (cases.config4, dict(n=1000), 128)])
def test_multiframe_edge_kernel_matrix(oracle, builder, kw, K):
    """BinaryStateGPU::update_inner_product = both frames under their own 3x4 pose, then
    fill_in_A_mat_gpu with the edge's K and ell, returned in the reference's host layout.  Bit-exact vs the oracle."""
    P, c1, c2, _ = builder(**kw)
    pose1 = _pose34(2.0, (0.1, 1.0, 0.2), (0.05, -0.02, 0.1))
    pose2 = _pose34(-1.0, (1.0, 0.2, 0.0), (-0.3, 0.05, -0.35))
    gpu = CvoGPU(params=P)
    f1, f2 = gpu.transformed(gpu.upload(c1), pose1), gpu.transformed(gpu.upload(c2), pose2)
    ell = 0.6
    mat, ind, nz, total = gpu.edge_kernel_matrix(f1, f2, ell, K)
    x1, fe1, la1, ge1 = c1.device_arrays()
    x2, fe2, la2, ge2 = c2.device_arrays()
    o1 = oracle.Cloud(oracle.transform_pose_vec(pose1, x1), fe1, la1, ge1)
    o2 = oracle.Cloud(oracle.transform_pose_vec(pose2, x2), fe2, la2, ge2)
    omat, oind, onz = oracle.se_kernel(oracle.params_from(P), o1, o2, K, ell)
    assert total == int(onz.sum()) and total > 0
    assert np.array_equal(nz, onz)
    assert np.array_equal(ind, oind)                  # ordered first-K truncation, -1 padding
    assert np.allclose(mat, omat, rtol=2e-6, atol=0)  # exp() is the only difference (ocml vs glibc, <= 1 ulp)


def test_binary_state_gpu_adapts_neighbours():
    """The reference's neighbour-count adaptation (IRLS_State_GPU.cu:45-47) around the same kernel."""
    from unified_cvo_amd import CvoFrameGPU, BinaryStateGPU
    P, c1, c2, _ = cases.config2(n=1200)
    gpu = CvoGPU(params=P)
    f1 = CvoFrameGPU(gpu, c1, _pose34(0.0, (0, 0, 1), (0, 0, 0)))
    f2 = CvoFrameGPU(gpu, c2, np.linalg.inv(np.vstack([synth.gt_motion()[:3], [0, 0, 0, 1]]))[:3])
    st = BinaryStateGPU(f1, f2, num_neighbor=128, init_ell=0.3)
    n0 = st.update_inner_product()
    assert n0 > 0 and st.mat.shape == (1200, 128)
    assert st.update_inner_product() == n0            # same frames, K shrinks to 1.1 * max row count but keeps every entry
    assert st.num_neighbors == min(128, int(int(st.nonzeros.max()) * 1.1))
    f2.pose_vec[3] += 0.4                            # move frame 2 and refresh it: the matrix changes
    f2.transform_pointcloud()
    assert st.update_inner_product() != n0


def _rot(angle_deg, axis):
    return _pose34(angle_deg, axis, (0, 0, 0))[:, :3]


@pytest.mark.parametrize("builder,kw", [(cases.config2, dict(n=1500)), (cases.config3, dict(n=1200)),
                                        (cases.config4, dict(n=1000))])
def test_non_isotropic_association(oracle, builder, kw):
    """compute_association_gpu(..., Matrix3f kernel) - Mahalanobis distance, no geometric cut-off,
    geometric types off.  The scan is steered by a Euclidean bound derived from sp_thres; the pattern must be bit-exact."""
    P, src, tgt, init = builder(**kw)
    Q = _rot(25.0, (0.3, 1.0, -0.2))
    kernel = (Q @ np.diag([0.05, 0.12, 0.02]) @ Q.T).astype(np.float32)
    T = synth.gt_motion().astype(np.float32)
    gpu = CvoGPU(params=P)
    rp, col, val = gpu.compute_association_gpu_non_isotropic(src, tgt, T, kernel)
    orp, ocol, oval, kinv = oracle.association_non_isotropic(oracle.params_from(P), _ocloud(oracle, src), _ocloud(oracle, tgt),
                                                             T, kernel)
    assert np.allclose(kinv @ kernel.astype(np.float64), np.eye(3), atol=1e-4)   # the restated Eigen inverse
    assert len(ocol) > 100
    assert np.array_equal(rp, orp)
    assert np.array_equal(col, ocol)
    assert np.allclose(val, oval, rtol=2e-6, atol=0)


def test_non_isotropic_association_without_a_bound(oracle):
    """An indefinite kernel has no Euclidean bound: every pair is a candidate, results still match."""
    P, src, tgt, init = cases.config2(n=400)
    kernel = np.diag([0.05, -0.08, 0.03]).astype(np.float32)
    gpu = CvoGPU(params=P)
    rp, col, val = gpu.compute_association_gpu_non_isotropic(src, tgt, np.eye(4, dtype=np.float32), kernel)
    orp, ocol, oval, _ = oracle.association_non_isotropic(oracle.params_from(P), _ocloud(oracle, src), _ocloud(oracle, tgt),
                                                          np.eye(4, dtype=np.float32), kernel)
    assert np.array_equal(rp, orp) and np.array_equal(col, ocol)
    assert np.allclose(val, oval, rtol=2e-6, atol=0)


def test_lidar_flavour_single_feature(oracle):
    """cvo_gpu_lidar_lib (FEATURE_DIMENSIONS = 1): clouds with one intensity channel; the backend pads features to
    five zeros, which leaves every colour distance unchanged."""
    P = cases.load_params("intensity_gpu")
    src, fsrc, tgt, ftgt, _, _ = synth.colour_pair(1500, 3)
    geo = np.tile(np.array([[0.0, 1.0]], np.float32), (1500, 1))
    a = CvoPointCloud.from_arrays(src, fsrc[:, :1], None, geo)
    b = CvoPointCloud.from_arrays(tgt, ftgt[:, :1], None, geo)
    assert a.num_features() == 1
    _single_iteration(oracle, P, a, b, np.eye(4, dtype=np.float32))
    g = CvoGPU(params=P).align(a, b, np.eye(4), max_iterations=60)
    o = oracle.align(oracle.params_from(P), _ocloud(oracle, a), _ocloud(oracle, b), np.eye(4, dtype=np.float32),
                     max_iterations=60)
    assert g.iterations == o["iterations"] == 60
    assert cases.max_abs_diff(g.transform, o["transform"]) < 1e-6


FUZZ_SEEDS = int(os.environ.get("CVO_FUZZ_SEEDS", "48"))
FUZZ_OUTCOMES = {}   # seed -> followed the oracle strictly to the last recorded iteration (filled by _fuzz_one)


@pytest.mark.parametrize("seed", range(FUZZ_SEEDS))
def test_randomised_trajectories(oracle, seed):
    """Randomised sizes / parameters / initial guesses: every recorded iteration (counts, ell, K, twist, B..E, step,
    pose) must follow the oracle through list rebuilds, waits, ordered truncation and the overflow path."""
    _fuzz_one(oracle, seed)


def _fuzz_one(oracle, seed):
    if seed in FUZZ_OUTCOMES:
        return FUZZ_OUTCOMES[seed]
    rs = np.random.default_rng(100 + seed)
    kind = seed % 3
    n = int(rs.integers(150, 1800))
    m = int(rs.integers(150, 1800))
    if seed % 8 == 7:  # a few at BASELINE scale (up to 10k x 10k, ragged)
        n = int(rs.integers(4000, 10001))
        m = int(rs.integers(4000, 10001))
    if kind == 0:
        P = cases.load_params("geometric_gpu")
        src, tgt, _ = synth.geometric_pair(n, seed, m=m)
        a, b = CvoPointCloud.from_xyz(src), CvoPointCloud.from_xyz(tgt)
    elif kind == 1:
        P = cases.load_params("intensity_gpu")
        src, fsrc, tgt, ftgt, _, _ = synth.colour_pair(max(n, m), seed)
        geo = np.tile(np.array([[0.0, 1.0]], np.float32), (max(n, m), 1))
        a = CvoPointCloud.from_arrays(src[:n], fsrc[:n], None, geo[:n])
        b = CvoPointCloud.from_arrays(tgt[:m], ftgt[:m], None, geo[:m])
    else:
        P = cases.load_params("semantic_img_gpu0")
        k = max(n, m)
        src, fsrc, lsrc, tgt, ftgt, ltgt = synth.semantic_pair(k, seed)
        geo = np.tile(np.array([[0.0, 1.0]], np.float32), (k, 1))
        a = CvoPointCloud.from_arrays(src[:n], fsrc[:n], lsrc[:n], geo[:n])
        b = CvoPointCloud.from_arrays(tgt[:m], ftgt[:m], ltgt[:m], geo[:m])
    P.ell_init = float(rs.choice([0.15, 0.3, 0.6, 1.2]))           # 1.2: dense, rows overflow their lists
    P.nearest_neighbors_max = int(rs.choice([8, 40, 512]))         # 8 / 40: the first-K truncation bites
    P.ell_decay_start = int(rs.choice([5, 30]))
    P.min_step = float(rs.choice([1e-4, 2e-3]))
    P.is_using_range_ell = int(rs.integers(0, 2))
    init = (synth.gt_motion() @ synth.warm_start_delta()).astype(np.float32) if rs.integers(0, 2) else np.eye(4, dtype=np.float32)
    strict = _follow_oracle(oracle, P, a, b, init, 90, seed)
    FUZZ_OUTCOMES[seed] = strict
    return strict


def _follow_oracle(oracle, P, a, b, init, n_it, seed):
    """A prefix of the optimisation, iteration by iteration against the oracle; returns whether the comparison stayed
    strict to the last recorded iteration (see the comment on the one-ulp separations below)."""
    g, o = _prefix(oracle, P, a, b, init, n_it)
    assert g.iterations == o["iterations"] and g.ret == o["ret"]
    assert len(g.trace) == len(o["trace"])
    compared = 0
    for x, y in zip(g.trace, o["trace"]):
        # exp() comes from two libms (ocml / glibc): once in ~10^6 kernel values the float result differs by one ulp,
        # which shows as a ~1e-8 relative difference of B..E and then separates the two trajectories (chaotically but
        # harmlessly).  Everything up to that iteration must agree strictly; the strict comparison stops there.
        if (x.k, x.K, x.nnz, x.max_nnz) == (y.k, y.K, y.nnz, y.max_nnz) and x.ell == y.ell and \
                any(1e-9 * max(abs(getattr(y, c)), 1e-12) + 1e-15 < abs(getattr(x, c) - getattr(y, c)) <= 1e-6 * max(abs(getattr(y, c)), 1e-12)
                    for c in "BCDE"):
            break
        _cmp_trace(x, y)
        compared += 1
    # (a separation in the very first iterations - sinf / cosf of Exp_SEK3 or exp() one ulp apart between ocml and glibc
    # right away: 2 of 400 clustered seeds - is not an agreement at all: such a pair is judged by where it ends, below)
    early = compared < min(10, len(g.trace))
    strict = compared == len(g.trace)
    if strict:
        assert cases.max_abs_diff(g.transform, o["transform"]) <= 1e-6
    else:
        # After a one-ulp separation the two runs are two valid trajectories of the same optimiser: the north_star
        # tolerance applies, relaxed to 2 * min_step where the prefix ends clamped at min_step (two
        # implementations can sit on opposite phases of the +-min_step jitter).
        # (the 2 * min_step allowance only where a run really ends on the clamp; 1e-3 at most otherwise)
        clamped = any(abs(t.trace[-1].step - P.min_step) <= 1e-6 * P.min_step for t in (g, _Obj(o)))
        tol = max(TOL_POSE_CLAMPED, 2.0 * P.min_step) if clamped else min(max(TOL_POSE_CLAMPED, 2.0 * P.min_step), 1e-3)
        if early or cases.max_abs_diff(g.transform, o["transform"]) > tol:
            # Two trajectories that separated early and are still far from the optimum when the prefix ends (seed 214 of
            # the clustered set: N = 2147 at ell = 1.4, every point a neighbour of every other, steps of 1e-2 at iteration
            # 70; identical to 1e-16 up to iteration 29, one exp() ulp at 30, 2.7e-3 apart at 70, 7e-6 apart at the end):
            # what has to agree then is where they END.
            gf = CvoGPU(params=P).align(a, b, init)
            of = oracle.align(oracle.params_from(P), _ocloud(oracle, a), _ocloud(oracle, b), init)
            assert (gf.iterations, gf.ret) == (of["iterations"], of["ret"]), (seed, compared)
            assert cases.max_abs_diff(gf.transform, of["transform"]) <= max(TOL_POSE_CLAMPED, 2.0 * P.min_step), (seed, compared, clamped)
    return strict


class _Obj:
    def __init__(self, d):
        self.trace = d["trace"]


@pytest.mark.parametrize("seed", range(int(os.environ.get("CVO_FUZZ_CLUSTERED", "12"))))
def test_randomised_clustered_trajectories(oracle, monkeypatch, seed):
    """The same for clustered clouds (synth.scene_pair: local density varies by more than 100x): random sizes, ragged
    pairs, lengthscales, neighbour caps and warm starts put rows on the 64-entry lists, on long lists, on the literal
    scan and on the K cap in one pair.  Every recorded iteration follows the oracle, and a second run re-derives every
    row of every iteration on the device (CVO_VERIFY_LISTS) and ends on the same pose."""
    rs = np.random.default_rng(900 + seed)
    n, m = int(rs.integers(1200, 6500)), int(rs.integers(1200, 6500))
    src, tgt, _ = synth.scene_pair(n, 50 + seed, m=m)
    kind = seed % 3   # geometry only / + colour / + colour and semantics (k_assoc, k_assoc_dense: GENERAL instantiations)
    if kind == 0:
        a, b = CvoPointCloud.from_xyz(src), CvoPointCloud.from_xyz(tgt)
        P = cases.load_params("geometric_gpu")
    else:
        T = synth.gt_motion()
        back = (tgt.astype(np.float64) - T[:3, 3]) @ T[:3, :3]          # the target points before the motion (+ noise)
        fs = synth.colour_features(src, np.random.default_rng(7100 + seed)).astype(np.float32)
        ft = synth.colour_features(back, np.random.default_rng(7200 + seed), noise=0.01).astype(np.float32)
        geo = np.tile(np.array([[0.0, 1.0]], np.float32), (max(n, m), 1))
        ls = synth.checkerboard_labels(src) if kind == 2 else None
        lt = synth.checkerboard_labels(back, flip=0.02, rng=np.random.default_rng(7300 + seed)) if kind == 2 else None
        a, b = CvoPointCloud.from_arrays(src, fs, ls, geo[:n]), CvoPointCloud.from_arrays(tgt, ft, lt, geo[:m])
        P = cases.load_params("intensity_gpu" if kind == 1 else "semantic_img_gpu0")
    P.ell_init = float(rs.choice([0.3, 0.6, 0.95, 1.4]))
    P.nearest_neighbors_max = int(rs.choice([40, 200, 512]))
    P.ell_decay_start = int(rs.choice([5, 30]))
    P.is_using_range_ell = int(rs.integers(0, 2))
    init = (synth.gt_motion() @ synth.warm_start_delta()).astype(np.float32) if rs.integers(0, 2) else np.eye(4, dtype=np.float32)
    n_it = 70
    _follow_oracle(oracle, P, a, b, init, n_it, seed)
    ref = CvoGPU(params=P).align(a, b, init, max_iterations=n_it)
    monkeypatch.setenv("CVO_VERIFY_LISTS", "1")
    gpu = CvoGPU(params=P)
    chk = gpu.align(a, b, init, max_iterations=n_it)          # raises CvoError(CVO_E_VERIFY) on any mismatch
    assert chk.iterations == ref.iterations and np.array_equal(chk.transform, ref.transform)
    assert gpu.debug_verified_rows() >= chk.iterations * a.num_points()


def test_randomised_trajectories_mostly_strict(oracle):
    """At most 10 % of the fuzz seeds may leave the strict per-iteration comparison (a one-ulp exp() difference between
    ocml and glibc); everything else must have followed the oracle bit-for-decision to the last recorded iteration.
    Self-contained: seeds the parametrised test has not run in this session (-k selections, xdist) are run here."""
    for seed in range(FUZZ_SEEDS):
        _fuzz_one(oracle, seed)
    loose = sorted(s for s, ok in FUZZ_OUTCOMES.items() if not ok)
    print(f"fuzz: {len(loose)} of {len(FUZZ_OUTCOMES)} seeds left the strict comparison: {loose}")
    assert len(loose) <= 0.10 * len(FUZZ_OUTCOMES), loose
