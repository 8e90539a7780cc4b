"""GPU tests of the API surface round 1 left untested, and of the device code pinned WITHOUT the oracle:

  * the 192-byte AoS route (pcl_PointCloud_to_gpu, CvoGPU_impl.cu:287-362) == the SoA route, bit for bit;
  * the Association align() exports (CvoGPU.cu:1552-1556 + CvoGPU_impl.cu:366-427): the LAST EXECUTED iteration's matrix;
  * the device's scalar maths (cubic, step selection, Exp_SEK3, ||SE3 log||, update_tf, indicator windows) against
    numpy / scipy directly (cvo_debug_scalar_math) - not against the oracle, which shares text with the device code;
  * CVO_VERIFY_LISTS=1: the device-side self-check of the candidate-list reuse at full size;
  * argument validation of the public override fields.
"""
import collections
import os

import numpy as np
import pytest
import scipy.linalg

import cases
import np_reference as npr
from unified_cvo_amd import (CvoGPU, CvoPointCloud, CvoError, synth, cvo_points_from_pointcloud, CVO_POINT_DTYPE)

pytestmark = pytest.mark.gpu


def _ocloud(oracle, pc):
    return oracle.Cloud.from_pointcloud(pc)


# ---------------------------------------------------------------------------------------------------------------
# 192-byte AoS records
# ---------------------------------------------------------------------------------------------------------------
def test_cvo_point_layout_is_192_bytes():
    assert CVO_POINT_DTYPE.itemsize == 192
    off = {n: CVO_POINT_DTYPE.fields[n][1] for n in CVO_POINT_DTYPE.names}
    assert (off["xyz"], off["rgba"], off["features"], off["label"], off["label_distribution"], off["geometric_type"],
            off["normal"], off["covariance"], off["cov_eigenvalues"]) == (0, 16, 20, 40, 44, 120, 128, 140, 176)


def test_aos192_upload_matches_soa_route_and_oracle(oracle):
    """pcl::PointCloud<CvoPoint> overloads: the AoS record IS the wire format.  Records are built in numpy at the
    offsets of PointSegmentedDistribution.hpp:17-99 (with garbage in every field the kernels must not read)."""
    P, src, tgt, init = cases.config4(n=2000)
    gpu = CvoGPU(params=P)
    rs = np.random.default_rng(3)
    recs = []
    for pc in (src, tgt):
        r = cvo_points_from_pointcloud(pc)
        r["pad_w"] = rs.normal(size=r.shape[0])        # never read
        r["normal"] = rs.normal(size=(r.shape[0], 3))
        r["covariance"] = rs.normal(size=(r.shape[0], 9))
        r["cov_eigenvalues"] = rs.normal(size=(r.shape[0], 3))
        r["label"] = rs.integers(0, 19, r.shape[0])
        recs.append(r)
    a_src, a_tgt = gpu.upload_aos192(recs[0]), gpu.upload_aos192(recs[1])
    n_it = 120
    g_aos = gpu.align(a_src, a_tgt, init, max_iterations=n_it, trace_capacity=n_it, trace_dense=n_it)
    g_soa = gpu.align(src, tgt, init, max_iterations=n_it, trace_capacity=n_it, trace_dense=n_it)
    assert g_aos.iterations == g_soa.iterations == n_it
    assert np.array_equal(g_aos.transform, g_soa.transform)
    for a, b in zip(g_aos.trace, g_soa.trace):
        assert (a.K, a.nnz, a.max_nnz, a.B, a.C, a.D, a.E) == (b.K, b.nnz, b.max_nnz, b.B, b.C, b.D, b.E)
    ip_a = gpu.inner_product_gpu(a_src, a_tgt, init, 0.2)
    assert ip_a == gpu.inner_product_gpu(src, tgt, init, 0.2)
    assert gpu.function_angle(a_src, a_tgt, init, 0.2, False) == gpu.function_angle(src, tgt, init, 0.2, False)
    o = oracle.align(oracle.params_from(P), _ocloud(oracle, src), _ocloud(oracle, tgt), init, max_iterations=n_it)
    assert cases.max_abs_diff(g_aos.transform, o["transform"]) <= 1e-6
    assert ip_a == pytest.approx(oracle.inner_product(oracle.params_from(P), _ocloud(oracle, src), _ocloud(oracle, tgt), init, 0.2),
                                 rel=1e-4)


# ---------------------------------------------------------------------------------------------------------------
# align(..., Association*)
# ---------------------------------------------------------------------------------------------------------------
def _assoc_triplets(row_ptr, col, val):
    rows = np.repeat(np.arange(len(row_ptr) - 1), np.diff(row_ptr))
    return rows.astype(np.int32), col, val


def _check_association(oracle, P, src, tgt, init, expect_same_stride=None):
    gpu = CvoGPU(params=P)
    g = gpu.align(src, tgt, init)
    rp, col, val, kw, kr = gpu.align_association(src.num_points())
    o = oracle.align_association(oracle.params_from(P), _ocloud(oracle, src), _ocloud(oracle, tgt), init)
    assert g.iterations == o["iterations"] and g.ret == o["ret"]
    assert (kw, kr) == (o["K_used"], o["K_final"])
    if expect_same_stride is not None:
        assert (kw == kr) == expect_same_stride
    rows, c, v = _assoc_triplets(rp, col, val)
    orow, ocol, oval = o["row"], o["col"], o["val"]
    if kr > kw:
        # The loop ran out of iterations right after num_neighbors GREW: upstream reads rows * kr entries of buffers
        # whose last iteration defined only the first rows * kw; what lies beyond are leftovers of earlier iterations
        # (the literal oracle keeps the buffers' history and returns them, the C-ABI ends the row there: DESIGN.md
        # "Association export").  Rows read entirely inside the defined part must still agree.
        n = src.num_points()
        inside = (orow.astype(np.int64) + 1) * kr <= n * kw
        assert not inside.all() or len(orow) == len(rows)
        gi = (rows.astype(np.int64) + 1) * kr <= n * kw
        orow, ocol, oval = orow[inside], ocol[inside], oval[inside]
        rows, c, v = rows[gi], c[gi], v[gi]
    assert np.array_equal(rows, orow) and np.array_equal(c, ocol)
    assert np.allclose(v, oval, rtol=2e-7, atol=0)
    if kw == kr:
        assert np.array_equal(np.nonzero(np.diff(rp) > 0)[0], o["source_inliers"])
    return g, o, (rp, col, val, kw, kr)


def test_align_association_after_eps2_break(oracle):
    """Config 4 stops through dist < eps_2: the exported matrix is the one of the last executed iteration (built at
    the pose BEFORE its update, with that iteration's ell and K) - not a fresh evaluation at the final pose."""
    P, src, tgt, init = cases.config4(n=2000)
    P.is_exporting_association = 1
    g, o, (rp, col, val, kw, kr) = _check_association(oracle, P, src, tgt, init, expect_same_stride=True)
    assert g.iterations < P.MAX_ITER and len(col) > 1000
    # and it is NOT what round 1 exported (re-evaluation at the final pose with K_max and the decayed ell)
    gpu = CvoGPU(params=P)
    rp2, col2, val2 = gpu.compute_association_gpu(src, tgt, np.linalg.inv(g.transform), g.final_ell)
    assert not (len(col2) == len(col) and np.array_equal(col2, col) and np.array_equal(val2, val))


def test_align_association_after_eps2_break_in_iteration_zero(oracle):
    """Warm start at the optimum with min_step < eps_2: the loop leaves through `dist < eps_2` in iteration 0, so
    `iterations == 0` and ret == 0 - and upstream still exports iteration 0's matrix (gpu_association_to_cpu runs
    after ANY loop, CvoGPU.cu:1505-1508, 1552), whose nonzero sum is > 0."""
    P, src, tgt, init = cases.config2(n=1500)
    warm = np.linalg.inv(CvoGPU(params=P).align(src, tgt, init).transform.astype(np.float64)).astype(np.float32)
    P.is_exporting_association = 1
    P.eps_2 = 1e-3                      # > min_step = 1e-4: the clamped first step already ends the loop
    g, o, (rp, col, val, kw, kr) = _check_association(oracle, P, src, tgt, warm, expect_same_stride=True)
    assert g.iterations == 0 and g.ret == 0
    assert kw == P.nearest_neighbors_max and len(col) > 1000


def test_align_association_after_max_iter(oracle):
    """The loop runs out of iterations: num_neighbors was already advanced for an iteration that never ran
    (CvoGPU.cu:1529) and upstream reads the buffers with that stride."""
    P, src, tgt, init = cases.config2(n=1500)
    P.is_exporting_association = 1
    P.MAX_ITER = 300
    _check_association(oracle, P, src, tgt, init)
    for max_iter in (1, 2, 3, 7):       # early cut-offs: K drops from K_max to 1.2 * max_nnz -> stride_read < stride_written
        P.MAX_ITER = max_iter
        g, o, (rp, col, val, kw, kr) = _check_association(oracle, P, src, tgt, init)
        if max_iter == 1:
            assert kw == P.nearest_neighbors_max and kr < kw


def test_align_association_empty(oracle):
    P, src, tgt, init = cases.config2(n=300)
    P.is_exporting_association = 1
    far = CvoPointCloud.from_xyz(tgt.positions() + np.float32(100.0))
    gpu = CvoGPU(params=P)
    g = gpu.align(src, far, init)
    rp, col, val, kw, kr = gpu.align_association(300)
    assert g.ret == -1 and len(col) == 0 and not rp.any()
    gpu.inner_product_gpu(src, tgt, init, 0.3)
    with pytest.raises(CvoError):          # the last call was not an align
        gpu.align_association(300)


# ---------------------------------------------------------------------------------------------------------------
# device scalar maths vs numpy / scipy
# ---------------------------------------------------------------------------------------------------------------
def _roots_match(dev, ref, tol):
    dev = sorted(dev, key=lambda z: (round(z.real, 9), z.imag))
    used = [False] * 3
    for z in dev:
        d = [abs(z - w) if not used[i] else np.inf for i, w in enumerate(ref)]
        i = int(np.argmin(d))
        if d[i] > tol * max(1.0, abs(ref[i])):
            return False
        used[i] = True
    return True


@pytest.mark.parametrize("op", [0, 1])
def test_device_cubic_roots_vs_numpy(op):
    """poly_solver_order3 (LieGroup.cpp:309-325: companion-matrix eigenvalues): both device variants - the scalar solver
    and the three-lane search the update uses - against numpy.roots (LAPACK)."""
    rs = np.random.default_rng(17)
    coefs = []
    for _ in range(300):
        s = 10.0 ** rs.uniform(-3, 3, 4)
        coefs.append(rs.normal(size=4) * s)
    for r in ([1, 2, 3], [-0.5, 1e-4, 40.0], [1e-5, 2e-5, 5.0], [2, 2, -1], [0.3, 0.3, 0.3]):   # real, incl. repeated
        coefs.append(np.poly(r) * rs.uniform(0.5, 2))
    for re, im in ((0.1, 2.0), (-3.0, 1e-3), (5.0, 40.0)):                                      # complex pairs
        coefs.append(np.real(np.poly([rs.normal(), complex(re, im), complex(re, -im)])))
    coefs = np.array(coefs)
    gpu = CvoGPU()
    out = gpu.debug_scalar_math(op, coefs)
    bad = 0
    for c, o in zip(coefs, out):
        dev = [complex(o[q], o[3 + q]) for q in range(3)]
        ref = np.roots(c)
        # relative residual of every device root (robust for ill-conditioned clusters), then root matching
        for z in dev:
            res = abs(np.polyval(c, z))
            scale = sum(abs(ck) * abs(z) ** (3 - k) for k, ck in enumerate(c))
            assert res <= 1e-9 * max(scale, 1e-300), (c, z, res, scale)
        bad += 0 if _roots_match(dev, ref, 1e-6) else 1
    assert bad <= 2          # (clustered roots: numpy itself is only good to ~sqrt(eps) there)
    # both device variants return identical roots
    assert np.array_equal(out[:, :6], gpu.debug_scalar_math(1 - op, coefs)[:, :6])


@pytest.mark.parametrize("op", [2, 3])
def test_device_step_selection_vs_numpy(op):
    """compute_step_size's host half (CvoGPU.cu:1122-1158): smallest root with positive real part and |imag| < 1e-5,
    clamped; "no admissible root -> max_step" (the overwrite quirk).  Includes the device's min_step shortcut."""
    rs = np.random.default_rng(23)
    items = []
    for _ in range(400):
        B, Cc, D, E = rs.normal(size=4) * 10.0 ** rs.uniform(-2, 5, 4)
        items.append([B, Cc, D, E, 10.0 ** rs.uniform(-6, -3), 10.0 ** rs.uniform(-2, 0)])
    # end-game shapes of the real loop: tiny positive root below min_step
    for _ in range(100):
        r0 = 10.0 ** rs.uniform(-7, -3)
        c = np.poly([r0, -abs(rs.normal()) - 0.05, abs(rs.normal()) + 0.06]) * rs.uniform(1e2, 1e6)
        items.append([c[3], c[2] / 2, c[1] / 3, c[0] / 4, 1e-4, 0.8])
    items.append([1.0, 1.0, 1.0, 1.0, 1e-4, 0.8])      # no positive root at all -> max_step
    items.append([-1.0, 0.0, 0.0, 0.0, 1e-4, 0.8])     # degenerate leading coefficient (NaN roots) -> max_step
    items = np.array(items)
    out = CvoGPU().debug_scalar_math(op, items)[:, 0]
    n_short = 0
    for it, s in zip(items, out):
        B, Cc, D, E, mn, mx = it
        mn32, mx32 = float(np.float32(mn)), float(np.float32(mx))
        with np.errstate(all="ignore"):
            ref = npr.step_from_coeffs(B, Cc, D, E, mn32, mx32) if E != 0 else mx32
        ref32 = float(np.float32(ref))
        if s == pytest.approx(ref32, rel=1e-5):
            n_short += s == np.float32(mn32)
            continue
        # a root within rounding of a clamp or of the |imag| < 1e-5 rule may legitimately fall either side
        r = np.roots([4 * E, 3 * D, 2 * Cc, B])
        near_rule = any(abs(abs(z.imag) - 1e-5) < 1e-7 or abs(z.real) < 1e-12 for z in r)
        assert near_rule, (it, s, ref32, r)
    assert n_short > 50


def _step_shortcut_items(rs, n_random):
    """Coefficient sets (B, C, D, E, min_step, max_step) around everything the certified Newton shortcut has to get right or
    refuse: the real loop's shapes (golden traces), small first roots with the other two far / near / almost equal, complex
    pairs in front of the first real root whose imaginary part straddles the 1e-5 rule, roots on the clamps and on float
    rounding boundaries, decreasing and increasing starts, both signs of B."""
    import json
    items = []
    d = json.load(open(os.path.join(cases.ROOT, "tests", "golden", "oracle_traces.json")))
    for c in d["cases"]:
        for t in c["trace"]:
            for mn, mx in ((1e-4, 0.8), (1e-6, 0.01), (1e-7, 0.01), (2e-5, 0.8)):
                items.append([t["B"], t["C"], t["D"], t["E"], mn, mx])

    def from_roots(roots, lead, mn, mx):
        c = np.real(np.poly(roots)) * lead          # 4E, 3D, 2C, B
        items.append([c[3], c[2] / 2, c[1] / 3, c[0] / 4, mn, mx])

    for _ in range(n_random):
        r0 = 10.0 ** rs.uniform(-7, 0)
        lead = rs.choice([-1, 1]) * 10.0 ** rs.uniform(0, 9)
        kind = rs.integers(0, 8)
        mn, mx = [(1e-4, 0.8), (1e-6, 0.01), (1e-7, 0.01)][rs.integers(0, 3)]
        if kind == 0:    # the other roots far away, either side
            from_roots([r0, rs.choice([-1, 1]) * r0 * 10 ** rs.uniform(0.3, 4), rs.choice([-1, 1]) * r0 * 10 ** rs.uniform(0.3, 4)], lead, mn, mx)
        elif kind == 1:  # a second root right behind the first (near-double root)
            from_roots([r0, r0 * (1 + 10.0 ** rs.uniform(-12, -1)), -rs.uniform(0.1, 3)], lead, mn, mx)
        elif kind == 2:  # complex pair in front of the first real root, |imag| around the 1e-5 rule
            u = r0 * rs.uniform(0.05, 0.95)
            from_roots([r0, complex(u, 10.0 ** rs.uniform(-7, -3)), complex(u, -(10.0 ** rs.uniform(-7, -3)))], lead, mn, mx)
            v = 10.0 ** rs.uniform(-7, -3)
            from_roots([r0, complex(u, v), complex(u, -v)], lead, mn, mx)
        elif kind == 3:  # complex pair of the size of the root (the first iterations of the BASELINE shapes)
            from_roots([r0, complex(-r0 * rs.uniform(0.5, 2), r0 * rs.uniform(0.5, 2)), 0], lead, mn, mx)
            items.pop()
            z = complex(-r0 * rs.uniform(0.5, 2), r0 * rs.uniform(0.5, 2))
            from_roots([r0, z, z.conjugate()], lead, mn, mx)
        elif kind == 4:  # the first root on a clamp or on a float rounding boundary, give or take a few ulps
            base = float(rs.choice([mn, mx, np.float32(r0)]))
            f = np.float32(base)
            mid = (float(f) + float(np.nextafter(f, np.float32(np.inf)))) / 2
            r = rs.choice([base, mid]) * (1 + rs.integers(-4, 5) * 2.0 ** -rs.integers(36, 53))
            from_roots([r, -rs.uniform(0.01, 3), rs.uniform(0.5, 30) + r], lead, mn, mx)
        elif kind == 5:  # no positive root / only roots beyond max_step
            from_roots([-r0, -rs.uniform(0.1, 3), rs.choice([-1.0, 1.0]) * (mx * rs.uniform(1.0, 50))], lead, mn, mx)
        elif kind == 6:  # increasing start: the first positive root lies behind a local maximum
            from_roots([-r0 * rs.uniform(0.1, 10), r0, r0 * rs.uniform(1.5, 100)], lead, mn, mx)
        else:            # anything
            B, Cc, D, E = rs.normal(size=4) * 10.0 ** rs.uniform(-2, 6, 4)
            items.append([B, Cc, D, E, mn, mx])
    return np.array(items)


def test_certified_newton_step_equals_the_full_solve_bit_for_bit():
    """select_step's shortcut for unclamped iterations (step_newton_certified, cvo_device.h) against the full bracketing solve
    on the device, on real-loop coefficients and on shapes built to sit on every edge of its certificate: identical floats,
    and the shortcut actually answers most real-loop cases (reference CvoGPU.cu:1136-1158,
    LieGroup.cpp:309-325)."""
    rs = np.random.default_rng(101)
    items = _step_shortcut_items(rs, 40000)
    gpu = CvoGPU()
    fast = gpu.debug_scalar_math(3, items)
    full = gpu.debug_scalar_math(13, items)[:, 0]
    diff = np.nonzero(fast[:, 0].view(np.uint64) != full.view(np.uint64))[0]
    assert diff.size == 0, [(items[i].tolist(), fast[i, 0], full[i]) for i in diff[:5]]
    path = fast[:, 1]
    n_real = sum(len(c["trace"]) for c in __import__("json").load(open(os.path.join(cases.ROOT, "tests", "golden", "oracle_traces.json")))["cases"]) * 4
    taken_real = np.count_nonzero(path[:n_real])
    unclamped_real = np.count_nonzero(fast[:n_real, 0] != items[:n_real, 4].astype(np.float32))
    # (the end-game shortcut answers the clamped cases first; of the rest the Newton shortcut must take nearly all)
    assert taken_real >= 0.9 * unclamped_real, (taken_real, unclamped_real, n_real)
    assert np.count_nonzero(path == 1) > 5000 and np.count_nonzero(path == 2) > 500 and np.count_nonzero(path == 0) > 500


def test_device_exp_sek3_vs_scipy():
    """Exp_SEK3 (LieGroup.cpp:244-274) == expm of the 4x4 twist times dt; theta < 1e-6 -> R = I, translation = v."""
    rs = np.random.default_rng(5)
    items = []
    for _ in range(200):
        xi = rs.normal(size=6)
        xi /= np.linalg.norm(xi)
        items.append(list(xi) + [10.0 ** rs.uniform(-5, 0)])
    items.append([0, 0, 0, 0.6, 0.0, 0.8, 0.01])           # pure translation: Jl = I, NOT dt * I
    items.append([1e-8, 0, 0, 0.6, 0.0, 0.8, 0.5])
    out = CvoGPU().debug_scalar_math(4, np.array(items))[:, :12].reshape(-1, 3, 4)
    for it, o in zip(items, out):
        w, v, dt = np.array(it[:3]), np.array(it[3:6]), it[6]
        if np.linalg.norm(np.float32(w)) < 1e-6:
            assert np.allclose(o[:, :3], np.eye(3)) and np.allclose(o[:, 3], np.float32(v))
            continue
        X = np.zeros((4, 4))
        X[:3, :3] = npr.hat(w)
        X[:3, 3] = v
        E = scipy.linalg.expm(X * dt)
        assert np.allclose(o[:, :3], E[:3, :3], atol=3e-7), (it, o, E)
        assert np.allclose(o[:, 3], E[:3, 3], atol=3e-7 + 2e-6 * dt)   # (1 - cos)/theta^2 in float at tiny dt * theta


def test_device_se3_log_norm_vs_scipy():
    """|| Sophus::SE3d(dRT).log() || (CvoGPU.cu:1473-1476) == norm of the (u, omega) coordinates of logm."""
    rs = np.random.default_rng(9)
    items = []
    for _ in range(200):
        w = rs.normal(size=3)
        w *= 10.0 ** rs.uniform(-7, 0.4) / np.linalg.norm(w)
        t = rs.normal(size=3) * 10.0 ** rs.uniform(-6, 0)
        R = scipy.linalg.expm(npr.hat(w))
        items.append(list(R.reshape(9)) + list(t))
    items.append(list(np.eye(3).reshape(9)) + [0.0, 0.0, 0.0])
    items.append(list(np.eye(3).reshape(9)) + [3e-5, -4e-5, 0.0])
    out = CvoGPU().debug_scalar_math(5, np.array(items))[:, 0]
    for it, o in zip(items, out):
        ref = npr._se3_log_norm(np.array(it[:9]).reshape(3, 3), np.array(it[9:12]))
        assert o == pytest.approx(ref, rel=1e-7, abs=1e-12), (it, o, ref)


def test_device_update_tf_vs_numpy():
    rs = np.random.default_rng(2)
    items = []
    for _ in range(50):
        R = scipy.linalg.expm(npr.hat(rs.normal(size=3))).astype(np.float32)
        items.append(list(R.reshape(9)) + list(rs.normal(size=3).astype(np.float32) * 5))
    out = CvoGPU().debug_scalar_math(6, np.array(items))
    for it, o in zip(items, out):
        R, T = np.array(it[:9]).reshape(3, 3), np.array(it[9:12])
        assert np.array_equal(o[:9].reshape(3, 3), R.T)              # transform = [R^T | -R^T T]  (CvoGPU.cu:94-112)
        assert np.allclose(o[9:12], -R.T @ T, atol=2e-6)


@pytest.mark.parametrize("window,thr", [(15, 0.2), (10, 0.001), (3, 0.05), (1, 0.5)])
def test_device_indicator_windows_vs_deque(window, thr):
    """The ring-buffer restatement of A_sparsity_indicator_ell_update that the update kernel runs, against the deque
    version written from the reference text (np_reference.IndicatorWindows): the literal three-`if` control flow."""
    rs = np.random.default_rng(window)
    seq = np.concatenate([rs.uniform(0.5, 1.5, 40), np.full(60, 1.0) + rs.normal(0, 1e-4, 60), rs.uniform(0.2, 3.0, 80),
                          np.linspace(2.0, 1.0, 120)]).astype(np.float32)
    dev = CvoGPU().debug_scalar_math(7, np.concatenate([[window, thr], seq.astype(np.float64)]))
    ref = npr.IndicatorWindows(window, thr)
    want = np.array([1.0 if ref.push(x) else 0.0 for x in seq])
    assert np.array_equal(dev, want)
    assert want.sum() >= 1


# ---------------------------------------------------------------------------------------------------------------
# CVO_VERIFY_LISTS: the list-reuse argument checked on the device at full size
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,builder,kw,n_it", [
    ("config2-10k", cases.config2, dict(n=10000), 320),
    ("config3", cases.config3, dict(n=10000), 320),
    ("config4", cases.config4, dict(n=10000), 320),
    ("config2-5k", cases.config2, dict(n=5000), 400),
    ("config1", cases.config1, {}, 1200),
    ("scene-10k", cases.scene, dict(n=10000), 400),     # clustered: thousands of rows over the list capacity / on the K cap
    ("scene-3k", cases.scene, dict(n=3000), 2000),
])
def test_verify_lists_full_size(monkeypatch, name, builder, kw, n_it):
    """Every row of every iteration - including the fast-moving first 30, where lists live for one or two iterations,
    and the lean graph's waits - re-derived with the literal scan and compared bit for bit on the device."""
    P, src, tgt, init = builder(**kw)
    ref = CvoGPU(params=P).align(src, tgt, init, max_iterations=n_it)
    monkeypatch.setenv("CVO_VERIFY_LISTS", "1")
    gpu = CvoGPU(params=P)
    g = gpu.align(src, tgt, init, max_iterations=n_it)      # raises CvoError(CVO_E_VERIFY) on any mismatch
    assert g.iterations == ref.iterations and np.array_equal(g.transform, ref.transform)
    assert gpu.debug_verified_rows() == g.iterations * src.num_points() + (0 if g.iterations == n_it else src.num_points())
    builds, iters, _ = gpu.debug_list_builds()
    assert builds < iters                                    # lists really were reused while being verified


def _blob_pair(n_blob=3000, n_bg=5000, seed=5):
    """A tight ball of n_blob points (every one of them sees the whole ball: far more candidates than a long list holds)
    in a sparse background of n_bg points, moved by the usual ground-truth motion."""
    from unified_cvo_amd import synth
    rs = np.random.default_rng(seed)
    blob = rs.normal(0.0, 0.05, (n_blob, 3)) + np.array([0.5, -0.3, 6.0])
    bg = np.stack([rs.uniform(-8, 8, n_bg), rs.uniform(-2, 3, n_bg), rs.uniform(2, 30, n_bg)], axis=1)
    pts = np.concatenate([blob, bg])[rs.permutation(n_blob + n_bg)]
    T = synth.gt_motion()
    tgt = (pts @ T[:3, :3].T + T[:3, 3] + rs.normal(0, 0.01, pts.shape))[rs.permutation(len(pts))]
    return CvoPointCloud.from_xyz(pts.astype(np.float32)), CvoPointCloud.from_xyz(tgt.astype(np.float32))


def test_long_lists_three_row_classes(monkeypatch):
    """One pair with all three kinds of rows at once - listed (<= 64 candidates), long-listed (<= 1024, the cached
    index-sorted lists of k_assoc_dense) and scanned literally (the 3000-point blob) - re-derived row by row on the device,
    and bit-identical to the same call with CVO_NO_LONG_LISTS=1 (every overflow row scanned, as in round 3)."""
    P = cases.load_params("geometric_gpu")
    src, tgt = _blob_pair()
    init = np.eye(4, dtype=np.float32)
    n_it = 60
    monkeypatch.setenv("CVO_VERIFY_LISTS", "1")
    gpu = CvoGPU(params=P)
    g = gpu.align(src, tgt, init, max_iterations=n_it)
    assert g.iterations == n_it and gpu.debug_verified_rows() == n_it * src.num_points()
    n_ovf, n_scan, dense = gpu.debug_row_classes()
    assert not dense and n_scan >= 2500 and n_ovf > n_scan          # blob rows are scanned, others walk long lists
    monkeypatch.delenv("CVO_VERIFY_LISTS")
    monkeypatch.setenv("CVO_NO_LONG_LISTS", "1")
    ref = CvoGPU(params=P)
    r = ref.align(src, tgt, init, max_iterations=n_it)
    assert np.array_equal(r.transform, g.transform) and (r.final_ell, r.final_num_neighbors) == (g.final_ell, g.final_num_neighbors)
    o2, s2, _ = ref.debug_row_classes()
    assert s2 == o2                                                   # without long lists every overflow row is scanned


def test_mixed_batch_with_and_without_overflow_rows(monkeypatch):
    """Sub-batches whose pairs ask for different graphs: clustered scene pairs (overflow rows: the graphs with
    k_assoc_dense, the heavy coefficient split) next to slab pairs (lean / calm graphs, one block per row block) and a
    small demo-like pair in the dense regime.  Every row of every iteration is re-derived on the device, and every pair
    ends bit-identical to the same pair solved alone - the split of a pair's coefficient reduction depends on its own
    state only."""
    P = cases.load_params("geometric_gpu")
    pairs = [cases.scene(n=3000, pair_id=0), cases.config2(n=3000, pair_id=1), cases.scene(n=3000, pair_id=2),
             cases.config2(n=3000, pair_id=3), cases.scene(n=2500, pair_id=4), cases.config2(n=2200, pair_id=5, m=3000),
             cases.scene(n=3000, pair_id=6), cases.config2(n=3000, pair_id=7), cases.scene(n=600, pair_id=8)]
    n_it = 300
    monkeypatch.setenv("CVO_VERIFY_LISTS", "1")
    gpu = CvoGPU(params=P)
    res = gpu.align_batch([p[1] for p in pairs], [p[2] for p in pairs], [p[3] for p in pairs], max_iterations=n_it)
    assert gpu.debug_last_geometry()[0] == 2                                     # two sub-batches
    assert gpu.debug_verified_rows() == sum(r.iterations * p[1].num_points() for r, p in zip(res, pairs))
    classes = [gpu.debug_row_classes(q) for q in range(len(pairs))]
    assert any(c[0] > 0 for c in classes) and any(c[0] == 0 for c in classes)    # both kinds really were in the batch
    monkeypatch.delenv("CVO_VERIFY_LISTS")
    solo = CvoGPU(params=P)
    for q, (p, r) in enumerate(zip(pairs, res)):
        one = solo.align(p[1], p[2], p[3], max_iterations=n_it)
        assert (one.iterations, one.ret, one.final_ell, one.final_num_neighbors) == (
            r.iterations, r.ret, r.final_ell, r.final_num_neighbors), q
        assert np.array_equal(one.transform, r.transform), q


def test_long_lists_survive_across_calls():
    """The cached long lists carry a generation tag (call serial, list build): a second call on the same context and
    workspace - same pair, then another pair of the same size - never reads a list of the call before."""
    P, src, tgt, init = cases.scene(n=3000)
    gpu = CvoGPU(params=P)
    a = gpu.align(src, tgt, init, max_iterations=120)
    b = gpu.align(src, tgt, init, max_iterations=120)
    assert np.array_equal(a.transform, b.transform)
    assert gpu.debug_row_classes()[0] > 0
    P2, src2, tgt2, _ = cases.scene(n=3000, pair_id=1)
    fresh = CvoGPU(params=P).align(src2, tgt2, init, max_iterations=120)
    reused = gpu.align(src2, tgt2, init, max_iterations=120)
    assert np.array_equal(fresh.transform, reused.transform)


def test_verify_lists_catches_a_broken_skin(monkeypatch):
    """The check is not vacuous: with the motion bound disabled (CVO_DEBUG_NO_MOTION_BOUND: lists are never rebuilt for
    motion) the first fast iterations lose pairs and the call fails with CVO_E_VERIFY."""
    P, src, tgt, init = cases.config2(n=3000)
    monkeypatch.setenv("CVO_VERIFY_LISTS", "1")
    monkeypatch.setenv("CVO_DEBUG_NO_MOTION_BOUND", "1")
    with pytest.raises(CvoError, match="CVO_VERIFY_LISTS"):
        CvoGPU(params=P).align(src, tgt, init, max_iterations=60)


@pytest.mark.parametrize("which", [1, 2])
def test_a_partial_that_never_arrives_ends_the_call_instead_of_hanging(monkeypatch, which):
    """Round 6: block partials cross blocks as data-tagged granules and the elected block POLLS for tags (cvo_wave.h).  The
    poll is bounded: with CVO_DEBUG_DROP_PARTIAL row block 1 of k_assoc (1) / k_coeff (2) never publishes, the flow gate /
    the update give up after PARTIAL_POLL_LIMIT polls, latch PairState::sync_err and the call returns CVO_E_HIP - within
    seconds, with the device alive (the next call on a fresh context is bit-identical to an undisturbed one)."""
    import time
    P, src, tgt, init = cases.config2(n=3000)
    good = CvoGPU(params=P).align(src, tgt, init, max_iterations=40)
    monkeypatch.setenv("CVO_DEBUG_DROP_PARTIAL", str(which))
    bad = CvoGPU(params=P)
    monkeypatch.delenv("CVO_DEBUG_DROP_PARTIAL")
    t0 = time.time()
    with pytest.raises(CvoError, match="never arrived"):
        bad.align(src, tgt, init, max_iterations=40)
    assert time.time() - t0 < 30.0
    again = CvoGPU(params=P).align(src, tgt, init, max_iterations=40)
    assert np.array_equal(again.transform, good.transform) and again.iterations == good.iterations


def test_verify_lists_batch(monkeypatch):
    monkeypatch.setenv("CVO_VERIFY_LISTS", "1")
    pairs = [cases.config2(n=3000, pair_id=p) for p in range(6)] + [cases.config2(n=1200, pair_id=9, m=2100)]
    gpu = CvoGPU(params=pairs[0][0])
    res = gpu.align_batch([p[1] for p in pairs], [p[2] for p in pairs], [p[3] for p in pairs], max_iterations=250)
    assert all(r.iterations == 250 for r in res)
    assert gpu.debug_verified_rows() == 250 * sum(p[1].num_points() for p in pairs)


# ---------------------------------------------------------------------------------------------------------------
# argument validation
# ---------------------------------------------------------------------------------------------------------------
def test_override_state_is_validated():
    """K0 beyond nearest_neighbors_max would write past the ELL: rejected, like a non-positive / NaN ell0."""
    P, src, tgt, init = cases.config2(n=300)
    P.nearest_neighbors_max = 16
    gpu = CvoGPU(params=P)
    for bad in (dict(K0=17), dict(K0=0), dict(K0=-3), dict(ell0=0.0), dict(ell0=-1.0), dict(ell0=float("nan")),
                dict(ell0=float("inf"))):
        with pytest.raises(CvoError):
            gpu.align(src, tgt, init, max_iterations=2, **bad)
    assert gpu.align(src, tgt, init, max_iterations=2, K0=16, ell0=0.3).iterations == 2


# ---- the hoisted arithmetic of the row loops (round 4): bit-for-bit against the plain forms, on the device ----

def _bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def _hoisted_pairs(gpu, op, operands, per_item):
    """operands: (n, per_item) doubles; returns (plain, hoisted) as flat arrays in operand order."""
    out = gpu.debug_scalar_math(op, operands)
    return out[:, 0:16:2].reshape(-1), out[:, 1:16:2].reshape(-1)


@pytest.mark.gpu
def test_hoisted_division_is_the_ieee_division_bit_for_bit():
    """exp(-d2 / (2.0 l l)) (CvoGPU.cu:552): the per-pair IEEE double division with its row-only part hoisted
    (rcp_refined once per row, div_by per candidate: 3 instructions instead of 12).  10^7 random operand pairs from the
    real ranges - d2 in [0, d2_thres], l in [ell_min, 1.2 ell_0] x the range factor - are divided both ways ON THE
    DEVICE and compared bit for bit; numpy's (correctly rounded) division is the third opinion."""
    gpu = CvoGPU()
    rs = np.random.default_rng(41)
    n_items = 1_250_000                                           # x 8 operand pairs = 10^7
    l = rs.uniform(0.02, 1.2 * 0.95 * 1.3, (n_items, 8)).astype(np.float32)      # lengthscales incl. range_ell
    den = 2.0 * l.astype(np.float64) * l.astype(np.float64)                       # (2.0 * l) * l, exact in double
    thr = (-2.0 * l.astype(np.float64) ** 2 * np.log(1e-3 / 0.01)).astype(np.float32)
    d2 = (rs.uniform(0, 1, (n_items, 8)) ** 2 * thr).astype(np.float32)           # squared distances below the cut-off
    num = -d2.astype(np.float64)
    ops = np.empty((n_items, 16))
    ops[:, 0::2] = num
    ops[:, 1::2] = den
    plain, hoisted = _hoisted_pairs(gpu, 8, ops, 8)
    nz = num.reshape(-1) != 0                                     # (-0 / d = -0 plain, +0 hoisted: exp() maps both to 1)
    assert np.array_equal(_bits(plain)[nz], _bits(hoisted)[nz])
    assert np.all(hoisted[~nz] == 0.0)
    assert np.array_equal(_bits(plain), _bits(num.reshape(-1) / den.reshape(-1)))  # the device division IS correctly rounded
    # edges of the float-derived operand space: tiniest / largest float numerators, extreme lengthscales
    edge_n = -np.array([1.4e-45, 1.1754944e-38, 1e-30, 1e-10, 1.0, 3.0e3, 1e20, 3.4e38], np.float32).astype(np.float64)
    for dd in (2.0 * 1e-8 ** 2, 2.0 * 1e-3 ** 2, 0.0098, 2.0, 2.0 * 5e3 ** 2, 2.0 * np.float64(np.float32(1e15)) ** 2):
        ops = np.zeros((1, 16))
        ops[0, 0::2] = edge_n
        ops[0, 1::2] = dd
        plain, hoisted = _hoisted_pairs(gpu, 8, ops, 8)
        assert np.array_equal(_bits(plain), _bits(hoisted)), dd
        assert np.array_equal(_bits(plain), _bits(edge_n / dd)), dd


@pytest.mark.gpu
def test_hoisted_division_by_six_is_the_ieee_division_bit_for_bit():
    """beta^3 / 6.0 (CvoGPU.cu:1072): float cubes promoted to double, divided by the literal 6.0 - with the reciprocal
    folded (Markstein correction) vs the 12-instruction IEEE sequence, 10^7 operands over 60 decades plus edges."""
    gpu = CvoGPU()
    rs = np.random.default_rng(43)
    n_items = 1_250_000
    beta = (rs.normal(size=(n_items, 8)) * 10.0 ** rs.uniform(-30, 12, (n_items, 8))).astype(np.float32)
    with np.errstate(over="ignore", under="ignore"):
        cube = (beta * beta * beta).astype(np.float64)            # float products, as the kernel forms them
    cube = np.where(np.isfinite(cube), cube, 1.0)
    ops = np.zeros((n_items, 16))
    ops[:, :8] = cube
    plain, hoisted = _hoisted_pairs(gpu, 9, ops, 8)
    nz = cube.reshape(-1) != 0
    assert np.array_equal(_bits(plain)[nz], _bits(hoisted)[nz])
    assert np.all(hoisted[~nz] == 0.0)
    assert np.array_equal(_bits(plain), _bits(cube.reshape(-1) / 6.0))
    edge = np.zeros((1, 16))
    edge[0, :8] = np.array([1.4e-45, -1.4e-45, 1.1754944e-38, 3.4028235e38, -3.4028235e38, 6.0, 1.0, -2.5e-20], np.float32)
    plain, hoisted = _hoisted_pairs(gpu, 9, edge, 8)
    assert np.array_equal(_bits(plain), _bits(hoisted))


@pytest.mark.gpu
@pytest.mark.parametrize("op", [10, 11])
def test_restated_exp_is_the_device_library_exp_bit_for_bit(op):
    """exp_ocml (the device library's double exp restated with three-operand FMAs and SGPR-resident constants;
    op 11: the variant for non-positive arguments, without the overflow clamp) against exp() itself on the device, bit
    for bit: 10^7 arguments where the kernels evaluate it (log(sp_thres / sigma^2) .. 0) plus the whole double range."""
    gpu = CvoGPU()
    rs = np.random.default_rng(47 + op)
    n_items = 1_250_000
    x = -rs.uniform(0, 1, (n_items, 8)) ** 2 * 12.0               # the kernels' range, dense near 0
    x[: n_items // 8] = -10.0 ** rs.uniform(-300, 3.1, (n_items // 8, 8))          # every magnitude down to underflow
    if op == 10:
        x[n_items // 8: n_items // 4] = rs.uniform(-760, 720, (n_items // 4 - n_items // 8, 8))   # incl. overflow / gradual underflow
    ops = np.zeros((n_items, 16))
    ops[:, :8] = x
    plain, hoisted = _hoisted_pairs(gpu, op, ops, 8)
    assert np.array_equal(_bits(plain), _bits(hoisted))
    edge = np.zeros((1, 16))
    edge[0, :8] = [0.0, -0.0, -745.2, -708.4, -1074.9, -1075.1, -np.inf, -1e308] if op == 11 else \
        [0.0, 709.7, 709.9, 1023.9, 1024.1, np.inf, -np.inf, np.nan]
    e_plain, e_hoisted = _hoisted_pairs(gpu, op, edge, 8)
    assert np.array_equal(_bits(e_plain), _bits(e_hoisted))
    # and the library function itself stays within 1 ulp of glibc's (numpy), the difference DESIGN.md section 4 allows
    xs = ops[:, :8].reshape(-1)
    sel = (xs > -700) & (xs < 700)
    ref = np.exp(xs[sel])
    assert np.max(np.abs(plain[sel] - ref) / np.spacing(ref)) <= 1.0


@pytest.mark.gpu
def test_hoisted_float_division_is_the_ieee_division_bit_for_bit():
    """omega_i / c, v_i / d (CvoGPU.cu:784-787): IEEE float divisions by a call-wide constant with the denominator's half
    hoisted (fdiv_prepare / fdiv_hoisted) - licensed per wave by fdiv_operands_safe, which this test also pins: wherever
    it accepts a numerator the hoisted quotient equals the plain one bit for bit (and numpy's), and it accepts every
    zero and every magnitude in [2^-100, 2^60)."""
    gpu = CvoGPU()
    rs = np.random.default_rng(53)
    n_items = 1_250_000
    num = (rs.normal(size=(n_items, 8)) * 10.0 ** rs.uniform(-44, 30, (n_items, 8))).astype(np.float32)
    num[:1000] = (rs.normal(size=(1000, 8)) * 10.0 ** rs.uniform(-3, 3, (1000, 8))).astype(np.float32)   # the flows' usual range
    num[1000:1010] = 0.0
    den = np.where(rs.uniform(size=(n_items, 8)) < 0.5, 7.0, 2.0 ** rs.uniform(-20, 20, (n_items, 8))).astype(np.float32)
    den *= np.where(rs.uniform(size=den.shape) < 0.2, -1, 1).astype(np.float32)
    ops = np.empty((n_items, 16))
    ops[:, 0::2] = num
    ops[:, 1::2] = den
    plain, hoisted = _hoisted_pairs(gpu, 12, ops, 8)
    plain32, hoisted32 = plain.astype(np.float32), hoisted.astype(np.float32)
    nf, df = num.reshape(-1), den.reshape(-1)
    with np.errstate(all="ignore"):
        ref = (nf / df).astype(np.float32)
    assert np.array_equal(plain32.view(np.uint32), ref.view(np.uint32))          # the device's plain division is IEEE
    refused = (hoisted == -1.0) & (plain != -1.0)
    mag = np.abs(nf.astype(np.float64))
    must_accept = (mag == 0) | ((mag >= 2.0 ** -100) & (mag < 2.0 ** 59))
    assert not np.any(refused & must_accept)
    assert np.all(refused[(mag != 0) & ((mag < 2.0 ** -101) | (mag >= 2.0 ** 61))])
    ok = ~refused & (nf != 0)
    assert ok.sum() > 4_000_000
    assert np.array_equal(plain32[ok].view(np.uint32), hoisted32[ok].view(np.uint32))
    assert np.all(hoisted32[~refused & (nf == 0)] == 0.0)


# ---- the spatial ordering computed on the device (k_kd_order, round 4) ----

def _leaf_sets(order):
    """Every aligned run of four sorted positions is one leaf of the k-d ordering: its point set, order-free."""
    n = len(order)
    pad = (-n) % 4
    o = np.concatenate([order, np.full(pad, -1, order.dtype)]).reshape(-1, 4)
    return np.sort(o, axis=1)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [9, 700, 4999, 10000, 16384])
def test_device_kd_ordering_equals_its_host_twin(n):
    """k_kd_order (one block per cloud: level-wise bitonic sorts of (segment, coordinate, point) keys in LDS) against
    the host recursion with the same rule (CVO_ORDER=virtual: std::nth_element, split axis from the root box halved per
    level): the same points in every leaf - every aligned run of 4 - hence in every aligned run of 64 and 512; inside a
    leaf the two leave their points in different (irrelevant) orders."""
    src, _, _ = synth.geometric_pair(n, 7)
    pc = CvoPointCloud.from_xyz(src)
    dev = CvoGPU()
    d = dev.upload(pc).debug_order()
    assert np.array_equal(np.sort(d), np.arange(n))
    host = CvoGPU()
    host.set_option("ORDER", "virtual")
    h = host.upload(pc).debug_order()
    assert np.array_equal(_leaf_sets(d), _leaf_sets(h))
    # and the aligned runs of 64 really are compact: their boxes are far smaller than the cloud's
    x = src[d][: (n // 64) * 64].reshape(-1, 64, 3)
    if len(x):
        run_vol = np.prod(x.max(axis=1) - x.min(axis=1), axis=1).mean()
        assert run_vol < 4.0 * (64.0 / n) * np.prod(src.max(axis=0) - src.min(axis=0))   # (a random 64-subset spans ~all of it)


@pytest.mark.gpu
def test_result_does_not_depend_on_where_the_ordering_runs():
    """Device ordering (default), host ordering with per-segment boxes (CVO_ORDER=host), its level-axes twin and no
    ordering at all: bit-identical poses, with colour and semantic attributes riding along (they are gathered into
    spatial order by the same kernel)."""
    for builder, kw in ((cases.config2, dict(n=3000)), (cases.config4, dict(n=2000))):
        P, src, tgt, init = builder(**kw)
        ref = CvoGPU(params=P).align(src, tgt, init, max_iterations=150)
        for opt, val in (("ORDER", "host"), ("ORDER", "virtual"), ("NO_SORT", "1")):
            g = CvoGPU(params=P)
            g.set_option(opt, val)
            r = g.align(src, tgt, init, max_iterations=150)
            assert r.iterations == ref.iterations and np.array_equal(r.transform, ref.transform), (opt, val)


@pytest.mark.gpu
def test_large_and_odd_clouds_are_ordered_on_the_host():
    """Above KD_MAX_POINTS (16384) and for non-finite coordinates the upload falls back to the host ordering / the
    identity: same API, valid permutations, usable clouds."""
    gpu = CvoGPU()
    src, tgt, _ = synth.geometric_pair(20000, 3)
    o = gpu.upload(CvoPointCloud.from_xyz(src)).debug_order()
    assert np.array_equal(np.sort(o), np.arange(20000))
    bad = src[:100].copy()
    bad[17, 1] = np.nan
    assert np.array_equal(gpu.upload(CvoPointCloud.from_xyz(bad)).debug_order(), np.arange(100))


@pytest.mark.gpu
def test_operands_outside_the_hoisted_divisions_domain_are_refused():
    """The row loops evaluate their IEEE divisions in a hoisted form that equals the plain one only while the
    denominators 2 l^2, 2 c_ell^2, 2 s_ell^2 stay far from zero / infinity (cvo_device.h, rcp_refined): lengthscales
    outside [1e-30, 1e15] and clouds with non-finite or astronomically large coordinates are rejected with CVO_E_INVALID
    instead of silently leaving that domain."""
    from unified_cvo_amd import CvoError
    P, src, tgt, init = cases.config2(n=600)
    gpu = CvoGPU(params=P)
    da, db = gpu.upload(src), gpu.upload(tgt)
    assert gpu.align(da, db, init, max_iterations=3).iterations == 3
    for field, bad in (("ell_min", 1e-38), ("ell_init", 0.0), ("ell_init", float("inf")), ("ell_min", float("nan"))):
        Pb = cases.load_params("geometric_gpu")
        setattr(Pb, field, bad)
        gb = CvoGPU(params=Pb)
        with pytest.raises(CvoError):
            gb.align(gb.upload(src), gb.upload(tgt), init, max_iterations=3)
    with pytest.raises(CvoError):
        gpu.inner_product_gpu(da, db, init, 1e-35)
    xyz = np.array(src.positions(), np.float32).copy()
    xyz[5, 2] = np.nan
    with pytest.raises(CvoError):
        gpu.align(gpu.upload(CvoPointCloud.from_xyz(xyz)), db, init, max_iterations=3)
    xyz[5, 2] = 3e20
    with pytest.raises(CvoError):
        gpu.align(da, gpu.upload(CvoPointCloud.from_xyz(xyz)), init, max_iterations=3)
    Pc = cases.load_params("intensity_gpu")
    Pc.c_ell = 1e-36
    P3, s3, t3, i3 = cases.config3(n=500)
    g3 = CvoGPU(params=Pc)
    with pytest.raises(CvoError):
        g3.align(g3.upload(s3), g3.upload(t3), i3, max_iterations=3)
    gpu.L.cvo_shutdown()   # (pooled streams of the destroyed contexts above: harmless to call at any time)
    assert gpu.align(da, db, init, max_iterations=3).iterations == 3

