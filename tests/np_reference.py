"""Independent float64 numpy re-derivation of ONE optimiser iteration of CvoGPU::align, written from
the mathematical description in SURVEY.md section 8(a) (rows K2, K3, K4, K5, H3), not from the
oracle's code.  It is a *second opinion* for the oracle: different language, different structure
(dense N x M matrices), no shared code.  Tolerances are therefore float32-level, not bitwise."""
import numpy as np


def kernel_matrix(P, x, y, fx, fy, lx, ly, gx, gy, K, ell, cache=None):
    """Dense N x M matrix of a_ij with the reference's cut-offs and first-K-per-row truncation.  `cache` (a dict the
    caller keeps across iterations) holds the pose-independent factors: geometric type, colour and semantic kernels."""
    n, m = x.shape[0], y.shape[0]
    x = x.astype(np.float64); y = y.astype(np.float64)
    sp = np.float64(np.float32(P.sp_thres))
    sigma2 = np.float64(np.float32(P.sigma)) ** 2
    geo = None
    if P.is_using_geometry:
        l = (np.linalg.norm(x, axis=1) / 500.0 + 1.0) * ell
        d2 = ((x[:, None, :] - y[None, :, :]) ** 2).sum(-1)
        thr = -2.0 * l * l * np.log(sp / sigma2)
        geo = (d2 < thr[:, None], sigma2 * np.exp(-d2 / (2.0 * l[:, None] ** 2)))
    if cache is not None and "A" in cache:
        A, keep = cache["A"], cache["keep"]
    else:
        A, keep = _static_factors(P, n, m, sp, fx, fy, lx, ly, gx, gy)
        if cache is not None:
            cache["A"], cache["keep"] = A, keep
    if geo is not None:
        keep = keep & geo[0]
        A = A * geo[1]
    with np.errstate(invalid="ignore"):
        keep = keep & (A > sp)
    # keep the first K qualifying j of every row, in ascending j
    rank = np.cumsum(keep, axis=1)
    keep &= rank <= K
    return np.where(keep, A, 0.0), keep


def _static_factors(P, n, m, sp, fx, fy, lx, ly, gx, gy):
    A = np.ones((n, m))
    keep = np.ones((n, m), bool)
    if P.is_using_geometric_type:
        na = (gx.astype(np.float64) ** 2).sum(1)[:, None]
        nb = (gy.astype(np.float64) ** 2).sum(1)[None, :]
        dot = gx.astype(np.float64) @ gy.astype(np.float64).T
        with np.errstate(invalid="ignore", divide="ignore"):
            gs = dot * dot / (na * nb)
        keep &= ~(gs < 0.01)
        A = A * gs
    if P.is_using_intensity:
        c2 = np.float64(np.float32(P.c_ell)) ** 2
        cs2 = np.float64(np.float32(P.c_sigma)) ** 2
        d2c = ((fx.astype(np.float64)[:, None, :] - fy.astype(np.float64)[None, :, :]) ** 2).sum(-1)
        keep &= d2c < -2.0 * c2 * np.log(sp / cs2)
        A = A * cs2 * np.exp(-d2c / (2.0 * c2))
    if P.is_using_semantics:
        se = np.float64(np.float32(P.s_ell)); ss2 = np.float64(np.float32(P.s_sigma)) ** 2
        d2s = ((lx.astype(np.float64)[:, None, :] - ly.astype(np.float64)[None, :, :]) ** 2).sum(-1)
        keep &= d2s < -2.0 * se * se * np.log(sp / ss2)
        A = A * ss2 * np.exp(-d2s / (2.0 * se * se))
    return A, keep


def hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=np.float64)


def iteration(P, x, y0, R, T, ell, K, fx=None, fy=None, lx=None, ly=None, gx=None, gy=None, cache=None):
    """Returns dict(nnz, max_nnz, omega, v, B, C, D, E) for state (R, T, ell, K)."""
    R = np.asarray(R, np.float64); T = np.asarray(T, np.float64)
    y = (y0.astype(np.float64) - T) @ R          # R^T (y0 - T), row-vector form
    xx = x.astype(np.float64)
    A, keep = kernel_matrix(P, xx, y, fx, fy, lx, ly, gx, gy, K, ell, cache)
    nnz_row = keep.sum(1)
    cross = np.cross(xx[:, None, :], y[None, :, :])
    omega = (A[:, :, None] * cross).sum((0, 1)) / np.float32(P.c)
    v = (A[:, :, None] * (y[None, :, :] - xx[:, None, :])).sum((0, 1)) / np.float32(P.d)
    xi = np.concatenate([omega, v])
    nrm = np.linalg.norm(xi)
    if nrm > 0:
        xi = xi / nrm
    omega, v = xi[:3], xi[3:]
    W = hat(omega)
    xiz = y @ W.T + v
    xi2z = y @ (W @ W).T + W @ v
    xi3z = y @ (W @ W @ W).T + W @ W @ v
    xi4z = y @ (W @ W @ W @ W).T + W @ W @ W @ v
    lrow = np.full(xx.shape[0], ell, np.float64)
    if P.is_using_range_ell:
        lrow = (np.linalg.norm(xx, axis=1) / 500.0 + 1.0) * ell
    tc = (1.0 / (2.0 * lrow * lrow))[:, None]
    diff = xx[:, None, :] - y[None, :, :]
    beta = -2.0 * tc * (xiz[None] * diff).sum(-1)
    gamma = -tc * ((xiz ** 2).sum(-1)[None] + 2.0 * (xi2z[None] * diff).sum(-1))
    delta = 2.0 * tc * (-(xiz * xi2z).sum(-1)[None] - (xi3z[None] * diff).sum(-1))
    eps = -tc * (((xi2z ** 2).sum(-1) + 2.0 * (xiz * xi3z).sum(-1))[None] + 2.0 * (xi4z[None] * diff).sum(-1))
    B = (A * beta).sum()
    C = (A * (gamma + beta ** 2 / 2)).sum()
    D = (A * (delta + beta * gamma + beta ** 3 / 6)).sum()
    E = (A * (eps + beta * delta + beta ** 2 * gamma / 2 + gamma ** 2 / 2 + beta ** 4 / 24)).sum()
    return dict(nnz=int(nnz_row.sum()), max_nnz=int(nnz_row.max()), omega=omega, v=v, B=B, C=C, D=D, E=E, A=A,
                keep=keep)


def step_from_coeffs(B, C, D, E, min_step, max_step):
    """Smallest positive real root (|imag| < 1e-5) of 4E s^3 + 3D s^2 + 2C s + B, clamped; no root -> max_step."""
    r = np.roots([4 * E, 3 * D, 2 * C, B])
    cand = [z.real for z in r if z.real > 0 and abs(z.imag) < 1e-5]
    if not cand:
        return max_step
    s = min(cand)
    return max_step if s > max_step else (min_step if s < min_step else s)


# ---------------------------------------------------------------------------------------------------------------
# The WHOLE loop of align_impl (CvoGPU.cu:1387-1531), again written from the reference text and sharing no code
# with oracle/: float64 dense numpy for the per-pair work, scipy expm / logm for the Lie-group steps, a deque for
# the indicator windows.  State that the reference STORES in float (R, T, ell, the indicator and its running sums)
# is rounded to float32 where the reference stores it, so that the discrete events of the loop - the iterations at
# which ell decays, the K sequence, the stop iteration - can be compared with the oracle's event for event.
# ---------------------------------------------------------------------------------------------------------------
import collections

f32 = np.float32


class IndicatorWindows:
    """A_sparsity_indicator_ell_update (CvoGPU.cu:1167-1285).  Three sequential `if`s - not else-if - so the sample
    that fills the start window is also the first sample of the end window, and the sample that fills the end window
    is judged at once and, if the ratio is not stable, enters the end window a second time."""

    def __init__(self, window, threshold):
        self.w, self.thr = int(window), f32(threshold)
        self.start, self.end = collections.deque(), collections.deque()
        self.start_sum, self.end_sum = f32(0), f32(0)

    def push(self, indicator):
        x = f32(indicator)
        decrease = False
        if len(self.start) < self.w:
            self.start.append(x)
            self.start_sum = f32(self.start_sum + x)
        if len(self.start) >= self.w and len(self.end) < self.w:
            self.end.append(x)
            self.end_sum = f32(self.end_sum + x)
        if len(self.start) >= self.w and len(self.end) >= self.w:
            ratio = f32(self.end_sum / self.start_sum) if self.start_sum != 0 else f32(np.inf)
            if f32(1) - self.thr < ratio < f32(1) + self.thr:
                decrease = True
                self.start.clear()
                self.end.clear()
                self.start_sum, self.end_sum = f32(0), f32(0)
            else:
                moved = self.end.popleft()
                self.end_sum = f32(self.end_sum - moved)
                self.start_sum = f32(self.start_sum + moved)
                self.start.append(moved)
                self.start_sum = f32(self.start_sum - self.start.popleft())
                self.end.append(x)
                self.end_sum = f32(self.end_sum + x)
        return decrease


def _twist_exp(omega, v, step):
    """Exp_SEK3 (LieGroup.cpp:244-274) as the matrix exponential of the 4x4 twist; the reference's theta < 1e-6
    branch returns R = I, Jl = I (translation v, NOT step * v)."""
    import scipy.linalg
    if np.linalg.norm(omega) < 1e-6:
        return np.eye(3), np.asarray(v, np.float64).copy()
    X = np.zeros((4, 4))
    X[:3, :3] = hat(omega)
    X[:3, 3] = v
    E = scipy.linalg.expm(X * float(step))
    return E[:3, :3], E[:3, 3]


def _se3_log_norm(dR, dT):
    """|| Sophus::SE3d(dRT).log() ||  (CvoGPU.cu:1473-1476): 6-vector (V^-1 t, theta * axis) of the matrix log."""
    import scipy.linalg
    M = np.eye(4)
    M[:3, :3] = dR
    M[:3, 3] = dT
    L = np.real(scipy.linalg.logm(M))
    w = np.array([L[2, 1], L[0, 2], L[1, 0]])
    return float(np.sqrt((L[:3, 3] ** 2).sum() + (w ** 2).sum()))


def align_loop(P, x, y0, init, max_iterations=0, fx=None, fy=None, lx=None, ly=None, gx=None, gy=None):
    """CvoGPU::align (CvoGPU.cu:1338-1632).  Returns dict(transform, ret, iterations, events) where events is one
    dict per executed iteration: k, K, ell, nnz, max_nnz, step, dist, decayed (bool)."""
    n, m = x.shape[0], y0.shape[0]
    if n == 0 or m == 0:
        return dict(transform=None, ret=0, iterations=0, events=[])
    R = np.asarray(init, np.float64)[:3, :3].astype(f32)          # 1363-1364
    T = np.asarray(init, np.float64)[:3, 3].astype(f32)
    ell = f32(P.ell_init)                                          # CvoState.cu:30
    K = int(P.nearest_neighbors_max)                               # 1385
    win = IndicatorWindows(P.indicator_window_size, P.indicator_stable_threshold)
    max_iter = int(P.MAX_ITER) if max_iterations <= 0 else min(int(P.MAX_ITER), int(max_iterations))
    ret, k, events = 0, 0, []
    cache = {}
    while k < max_iter:                                            # 1387
        it = iteration(P, x, y0, R.astype(np.float64), T.astype(np.float64), float(ell), K, fx, fy, lx, ly, gx, gy, cache)
        omega, v = it["omega"].astype(f32), it["v"].astype(f32)    # the twist is uploaded / read back as float
        step = step_from_coeffs(it["B"], it["C"], it["D"], it["E"], float(f32(P.min_step)), float(f32(P.max_step)))
        step = f32(step)
        ev = dict(k=k, K=K, ell=float(ell), nnz=it["nnz"], max_nnz=it["max_nnz"], step=float(step), dist=0.0, decayed=False)
        events.append(ev)
        if np.linalg.norm(omega.astype(np.float64)) < P.eps and np.linalg.norm(v.astype(np.float64)) < P.eps:  # 1454-1458
            if np.linalg.norm(omega) < 1e-8 and np.linalg.norm(v) < 1e-8:
                ret = -1
            break
        dR, dT = _twist_exp(omega.astype(np.float64), v.astype(np.float64), step)   # 1462
        dR, dT = dR.astype(f32).astype(np.float64), dT.astype(f32).astype(np.float64)  # dtrans is a float matrix
        T = (R.astype(np.float64) @ dT + T.astype(np.float64)).astype(f32)          # 1466
        R = (R.astype(np.float64) @ dR).astype(f32)                                  # 1469
        dist = _se3_log_norm(dR, dT)                                                 # 1473-1476
        ev["dist"] = dist
        ip_curr = f32(float(it["nnz"]) / np.sqrt(float(n) * float(m)))               # 1486
        need_decay = win.push(ip_curr)                                               # 1487-1492
        if dist < P.eps_2:                                                           # 1505-1508
            break
        if k > P.ell_decay_start and need_decay:                                     # 1509-1513
            ell = f32(ell * f32(P.ell_decay_rate))
            if ell < f32(P.ell_min):
                ell = f32(P.ell_min)
            ev["decayed"] = True
        K = min(int(P.nearest_neighbors_max), int(it["max_nnz"] * 1.2))              # 1529
        k += 1
    Rt = R.astype(np.float64).T                                                      # update_tf, 94-112 / 1562
    out = np.eye(4)
    out[:3, :3] = Rt
    out[:3, 3] = -Rt @ T.astype(np.float64)
    return dict(transform=out, ret=ret, iterations=k, events=events)


def inner_product_cpu(P, x, y, T_t2s, ell, fx=None, fy=None, lx=None, ly=None):
    """CvoGPU::inner_product_cpu (upstream src/cvo/CvoGPU.cpp:95-213), dense: moving points under T^-1, every pair with
    squared distance below d2_thres = -2 l^2 log(sp_thres / sigma^2) (the nanoflann radius search), plain ell, no
    neighbour cap, NO cut-off on the colour / semantic kernels, a = ck k sk kept if a > sp_thres; returns sum(a)."""
    x = x.astype(np.float64)
    Ti = np.linalg.inv(np.asarray(T_t2s, np.float64))
    ym = y.astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]
    sp = float(np.float32(P.sp_thres))
    s2 = float(np.float32(P.sigma)) ** 2
    d2 = ((x[:, None, :] - ym[None, :, :]) ** 2).sum(-1)
    near = d2 < -2.0 * ell * ell * np.log(sp / s2)
    A = np.ones_like(d2)
    if P.is_using_semantics:
        d2s = ((lx.astype(np.float64)[:, None, :] - ly.astype(np.float64)[None, :, :]) ** 2).sum(-1)
        A = A * float(np.float32(P.s_sigma)) ** 2 * np.exp(-d2s / (2.0 * float(np.float32(P.s_ell)) ** 2))
    if P.is_using_geometry:
        A = A * s2 * np.exp(-d2 / (2.0 * ell * ell))
    if P.is_using_intensity:
        d2c = ((fx.astype(np.float64)[:, None, :] - fy.astype(np.float64)[None, :, :]) ** 2).sum(-1)
        A = A * float(np.float32(P.c_sigma)) ** 2 * np.exp(-d2c / (2.0 * float(np.float32(P.c_ell)) ** 2))
    return float(A[near & (A > sp)].sum())
