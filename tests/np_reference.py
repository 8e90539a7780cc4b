"""Independent float64 numpy re-derivation of ONE optimiser iteration of CvoGPU::align, written from
the mathematical description in SURVEY.md section 8(a) (rows K2, K3, K4, K5, H3), not from the
oracle's code.  It is a *second opinion* for the oracle: different language, different structure
(dense N x M matrices), no shared code.  Tolerances are therefore float32-level, not bitwise."""
import numpy as np


def kernel_matrix(P, x, y, fx, fy, lx, ly, gx, gy, K, ell):
    """Dense N x M matrix of a_ij with the reference's cut-offs and first-K-per-row truncation."""
    n, m = x.shape[0], y.shape[0]
    x = x.astype(np.float64); y = y.astype(np.float64)
    sp = np.float64(np.float32(P.sp_thres))
    sigma2 = np.float64(np.float32(P.sigma)) ** 2
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
    if P.is_using_geometry:
        l = (np.linalg.norm(x, axis=1) / 500.0 + 1.0) * ell
        d2 = ((x[:, None, :] - y[None, :, :]) ** 2).sum(-1)
        thr = -2.0 * l * l * np.log(sp / sigma2)
        keep &= d2 < thr[:, None]
        A = A * sigma2 * np.exp(-d2 / (2.0 * l[:, None] ** 2))
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
    with np.errstate(invalid="ignore"):
        keep &= A > sp
    # keep the first K qualifying j of every row, in ascending j
    rank = np.cumsum(keep, axis=1)
    keep &= rank <= K
    return np.where(keep, A, 0.0), keep


def hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=np.float64)


def iteration(P, x, y0, R, T, ell, K, fx=None, fy=None, lx=None, ly=None, gx=None, gy=None):
    """Returns dict(nnz, max_nnz, omega, v, B, C, D, E) for state (R, T, ell, K)."""
    R = np.asarray(R, np.float64); T = np.asarray(T, np.float64)
    y = (y0.astype(np.float64) - T) @ R          # R^T (y0 - T), row-vector form
    xx = x.astype(np.float64)
    A, keep = kernel_matrix(P, xx, y, fx, fy, lx, ly, gx, gy, K, ell)
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
