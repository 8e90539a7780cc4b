"""Pins for the oracle's scalar building blocks against numpy / scipy (the third-party arithmetic of
the reference -- Eigen's eigenvalues(), Sophus' SE3d::log(), Eigen normalize() -- is not in the
repository, so these independent implementations are the anchors)."""
import collections

import numpy as np
import pytest
import scipy.linalg

from np_reference import hat, step_from_coeffs


def _sorted_roots(z):
    z = np.asarray(z, np.complex128)
    return z[np.lexsort((z.imag, z.real))]


@pytest.mark.parametrize("seed", range(8))
def test_cubic_roots_random_vs_numpy(oracle, seed):
    rs = np.random.default_rng(seed)
    for _ in range(200):
        c = rs.normal(size=4) * 10.0 ** rs.integers(-3, 4, size=4)
        if abs(c[0]) < 1e-6:
            continue
        a = _sorted_roots(oracle.cubic_roots(c))
        b = _sorted_roots(np.roots(c))
        scale = max(1.0, np.max(np.abs(b)))
        assert np.max(np.abs(a - b)) <= 1e-7 * scale, (c, a, b)


def test_cubic_roots_structured(oracle):
    for roots in ([1, 2, 3], [0.001, 5, -7], [1e-4, 1e-4 + 1e-3, 4], [2, 2, -1], [-1, -2, -3]):
        c = np.poly(roots)
        a = np.sort(oracle.cubic_roots(c).real)
        assert np.allclose(a, np.sort(roots), rtol=1e-6, atol=1e-6), (roots, a)
    # one real + complex pair
    c = np.poly([0.3, 1 + 2j, 1 - 2j]).real
    r = oracle.cubic_roots(c)
    assert np.allclose(_sorted_roots(r), _sorted_roots(np.roots(c)), atol=1e-10)


def test_cubic_degenerate_leading_zero_is_nan(oracle):
    r = oracle.cubic_roots([0.0, 1.0, 2.0, 3.0])
    assert np.all(np.isnan(r.real))


@pytest.mark.parametrize("seed", range(4))
def test_select_step_vs_numpy(oracle, seed):
    rs = np.random.default_rng(100 + seed)
    for _ in range(300):
        B, C, D, E = rs.normal(size=4) * 10.0 ** rs.integers(-2, 5, size=4)
        got = oracle.select_step(B, C, D, E, 1e-4, 0.8)
        want = step_from_coeffs(B, C, D, E, 1e-4, 0.8)
        assert abs(got - np.float32(want)) <= 1e-6 * max(1.0, abs(want)), (B, C, D, E, got, want)


def test_select_step_quirks(oracle):
    # no admissible root -> temp_step stays DBL_MAX -> the `> max_step` branch wins (CvoGPU.cu:1151-1158)
    assert oracle.select_step(1.0, 1.0, 1.0, 1.0, 1e-4, 0.8) == pytest.approx(0.8)
    # all coefficients zero (empty A): 0/0 -> NaN roots -> max_step
    assert oracle.select_step(0.0, 0.0, 0.0, 0.0, 1e-4, 0.8) == pytest.approx(0.8)
    # tiny positive root -> clamped to min_step
    c = np.poly([1e-7, -3.0, -5.0])  # monic cubic with smallest positive root 1e-7
    E, D, C, B = c[0] / 4, c[1] / 3, c[2] / 2, c[3]
    assert oracle.select_step(B, C, D, E, 1e-4, 0.8) == pytest.approx(1e-4)


@pytest.mark.parametrize("seed", range(5))
def test_exp_sek3_vs_expm(oracle, seed):
    rs = np.random.default_rng(seed)
    xi = rs.normal(size=6)
    xi /= np.linalg.norm(xi)
    for dt in (1e-4, 0.01, 0.3, 0.8):
        got = oracle.exp_sek3(xi, dt)
        X = np.zeros((4, 4))
        X[:3, :3] = hat(xi[:3])
        X[:3, 3] = xi[3:]
        want = scipy.linalg.expm(dt * X)[:3, :]
        assert np.allclose(got, want, atol=3e-7), (dt, got - want)


def test_exp_sek3_small_theta_quirk(oracle):
    # theta < 1e-6: R = I and Jl = I (NOT dt*I), LieGroup.cpp:252-255
    xi = np.array([1e-8, 0, 0, 0.6, 0.8, 0.0], np.float32)
    got = oracle.exp_sek3(xi, 0.01)
    assert np.array_equal(got[:, :3], np.eye(3, dtype=np.float32))
    assert np.allclose(got[:, 3], xi[3:])


@pytest.mark.parametrize("seed", range(6))
def test_se3_log_norm_vs_logm(oracle, seed):
    rs = np.random.default_rng(seed)
    for scale in (1e-6, 1e-3, 0.1, 1.0):
        xi = rs.normal(size=6) * scale
        X = np.zeros((4, 4))
        X[:3, :3] = hat(xi[:3])
        X[:3, 3] = xi[3:]
        Tm = scipy.linalg.expm(X)
        got = oracle.se3_log_norm(Tm[:3, :3], Tm[:3, 3])
        assert got == pytest.approx(np.linalg.norm(xi), rel=1e-8, abs=1e-14)


def test_se3_log_norm_identity(oracle):
    assert oracle.se3_log_norm(np.eye(3), np.zeros(3)) == 0.0


def _indicator_reference(seq, window, thr):
    """Independent restatement with deques, from the prose of SURVEY.md row H7."""
    start, end = collections.deque(), collections.deque()
    ssum = esum = np.float32(0)
    out = []
    for x in seq:
        x = np.float32(x)
        dec = False
        if len(start) < window:
            start.append(x); ssum = np.float32(ssum + x)
        if len(start) >= window and len(end) < window:
            end.append(x); esum = np.float32(esum + x)
        if len(start) >= window and len(end) >= window:
            ratio = np.float32(esum / ssum)
            if ratio > np.float32(1) - np.float32(thr) and ratio < np.float32(1) + np.float32(thr):
                dec = True
                start.clear(); end.clear()
                ssum = esum = np.float32(0)
            else:
                f = end.popleft()
                esum = np.float32(esum - f); ssum = np.float32(ssum + f)
                start.append(f)
                g = start.popleft()
                ssum = np.float32(ssum - g)
                end.append(x); esum = np.float32(esum + x)
        out.append(dec)
    return np.array(out)


@pytest.mark.parametrize("window,thr", [(3, 0.02), (10, 0.001), (15, 0.02), (30, 0.2)])
def test_indicator_sequence(oracle, window, thr):
    rs = np.random.default_rng(window)
    base = np.concatenate([np.linspace(8, 3, 200), np.full(150, 3.0), np.linspace(3, 1, 150)])
    seq = (base * (1 + 0.01 * rs.normal(size=base.size))).astype(np.float32)
    with np.errstate(all="ignore"):
        want = _indicator_reference(seq, window, thr)
    got = oracle.indicator_run(seq, window, thr)
    assert np.array_equal(got, want)
    assert want.any()


def test_update_tf_and_transform(oracle):
    rs = np.random.default_rng(3)
    A = rs.normal(size=(3, 3))
    R, _ = np.linalg.qr(A)
    T = rs.normal(size=3)
    y0 = rs.normal(size=(50, 3)) * 10
    Ri, Ti, yt = oracle.transform_cloud(R, T, y0)
    assert np.allclose(Ri, R.T, atol=1e-7)
    assert np.allclose(Ti, -R.T @ T, atol=1e-6)
    assert np.allclose(yt, (y0 - T) @ R, atol=1e-5)
