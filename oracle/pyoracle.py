"""ctypes wrapper of the CPU parity oracle (oracle/libcvo_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under unified_cvo_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libcvo_oracle.so")

FD, NC = 5, 19


class OracleParams(C.Structure):
    _fields_ = [
        ("ell_init", C.c_float), ("ell_min", C.c_float), ("sigma", C.c_float), ("sp_thres", C.c_float),
        ("c", C.c_float), ("d", C.c_float), ("c_ell", C.c_float), ("c_sigma", C.c_float), ("s_ell", C.c_float),
        ("s_sigma", C.c_float), ("MAX_ITER", C.c_int), ("eps", C.c_float), ("eps_2", C.c_float),
        ("min_step", C.c_float), ("max_step", C.c_float), ("nearest_neighbors_max", C.c_int),
        ("ell_decay_rate", C.c_float), ("ell_decay_start", C.c_int), ("indicator_window_size", C.c_int),
        ("indicator_stable_threshold", C.c_float), ("is_using_geometry", C.c_int),
        ("is_using_intensity", C.c_int), ("is_using_semantics", C.c_int), ("is_using_range_ell", C.c_int),
        ("is_using_geometric_type", C.c_int),
    ]


class OracleCloud(C.Structure):
    _fields_ = [("n", C.c_int), ("xyz", C.POINTER(C.c_float)), ("feat", C.POINTER(C.c_float)),
                ("label", C.POINTER(C.c_float)), ("geo", C.POINTER(C.c_float))]


class OracleTrace(C.Structure):
    _fields_ = [
        ("k", C.c_int), ("K", C.c_int), ("ell", C.c_float), ("step", C.c_float), ("nnz", C.c_uint),
        ("max_nnz", C.c_uint), ("omega", C.c_float * 3), ("v", C.c_float * 3), ("B", C.c_double),
        ("C", C.c_double), ("D", C.c_double), ("E", C.c_double), ("dist", C.c_double), ("R", C.c_float * 9),
        ("T", C.c_float * 3),
    ]


def build(force=False):
    src = [os.path.join(HERE, f) for f in ("cvo_oracle.cpp", "cvo_oracle.h", "Makefile")]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src):
        subprocess.check_call(["make", "-C", HERE, "-s"] + (["-B"] if force else []))
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    _lib = _bind(C.CDLL(LIB_PATH))
    return _lib


VARIANT_FMA = ("dev", "none", "all")
VARIANT_SUM = ("def", "alt")


def build_variants():
    """The same source under the other plausible floating-point conventions (oracle/Makefile, target `variants`)."""
    subprocess.check_call(["make", "-C", HERE, "-s", "-j4", "variants"])


class variant:
    """Context manager: every pyoracle call inside runs on the oracle built with the given conventions.
    fma: 'dev' (default build) | 'none' | 'all';  sum3: 'def' | 'alt'  (see oracle/Makefile)."""
    _cache = {}

    def __init__(self, fma="dev", sum3="def"):
        assert fma in VARIANT_FMA and sum3 in VARIANT_SUM
        self.key = f"{fma}_{sum3}"

    def __enter__(self):
        global _lib
        if self.key not in variant._cache:
            path = os.path.join(HERE, "_variants", f"libcvo_oracle_{self.key}.so")
            srcs = [os.path.join(HERE, f) for f in ("cvo_oracle.cpp", "cvo_oracle.h", "Makefile")]
            if not os.path.exists(path) or any(os.path.getmtime(q) > os.path.getmtime(path) for q in srcs):
                build_variants()
            variant._cache[self.key] = _bind(C.CDLL(path))
        self.prev = lib()
        _lib = variant._cache[self.key]
        return self

    def __exit__(self, *exc):
        global _lib
        _lib = self.prev
        return False


class OracleAssociation(C.Structure):
    _fields_ = [("cap_pairs", C.c_int), ("cap_rows", C.c_int), ("row", C.POINTER(C.c_int)), ("col", C.POINTER(C.c_int)),
                ("val", C.POINTER(C.c_float)), ("source_inliers", C.POINTER(C.c_int)), ("n_pairs", C.c_int),
                ("n_source_inliers", C.c_int), ("K_used", C.c_int), ("K_final", C.c_int), ("overflow", C.c_int)]


def _bind(L):
    dp, fp, ip = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int)
    pp, cp, tp = C.POINTER(OracleParams), C.POINTER(OracleCloud), C.POINTER(OracleTrace)
    L.oracle_cubic_roots.argtypes = [dp, dp, dp]
    L.oracle_cubic_roots.restype = None
    L.oracle_select_step.argtypes = [C.c_double] * 4 + [C.c_float, C.c_float]
    L.oracle_select_step.restype = C.c_float
    L.oracle_exp_sek3.argtypes = [fp, C.c_float, fp]
    L.oracle_exp_sek3.restype = None
    L.oracle_se3_log_norm.argtypes = [dp, dp]
    L.oracle_se3_log_norm.restype = C.c_double
    L.oracle_indicator_run.argtypes = [fp, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_ubyte)]
    L.oracle_indicator_run.restype = None
    L.oracle_update_tf.argtypes = [fp, fp, fp, fp]
    L.oracle_update_tf.restype = None
    L.oracle_transform.argtypes = [fp, fp, C.c_int, fp, fp]
    L.oracle_association_non_isotropic.restype = C.c_int
    L.oracle_association_non_isotropic.argtypes = [pp, cp, cp, fp, fp, ip, ip, fp, fp]
    L.oracle_transform_pose_vec.restype = None
    L.oracle_transform_pose_vec.argtypes = [fp, C.c_int, fp, fp]
    L.oracle_transform.restype = None
    L.oracle_se_kernel.argtypes = [pp, cp, cp, C.c_int, C.c_float, fp, ip, C.POINTER(C.c_uint), C.c_int]
    L.oracle_se_kernel.restype = None
    L.oracle_iteration.argtypes = [pp, cp, cp, fp, fp, C.c_float, C.c_int, tp, ip, fp, ip, C.POINTER(C.c_uint)]
    L.oracle_iteration.restype = C.c_int
    L.oracle_align.argtypes = [pp, cp, cp, fp, fp, ip, tp, C.c_int, C.c_int, C.c_int, ip, dp, C.c_int]
    L.oracle_align.restype = C.c_int
    L.oracle_inner_product.argtypes = [pp, cp, cp, fp, C.c_float]
    L.oracle_inner_product.restype = C.c_float
    L.oracle_function_angle.argtypes = [pp, cp, cp, fp, C.c_float, C.c_int]
    L.oracle_function_angle.restype = C.c_float
    L.oracle_association.argtypes = [pp, cp, cp, fp, C.c_float, ip, ip, fp]
    L.oracle_association.restype = C.c_int
    L.oracle_set_grid.argtypes = [C.c_int]
    L.oracle_set_grid.restype = None
    L.oracle_scan_seconds.argtypes = [C.c_int]
    L.oracle_scan_seconds.restype = C.c_double
    L.oracle_num_threads.restype = C.c_int
    L.oracle_set_num_threads.argtypes = [C.c_int]
    L.oracle_set_num_threads.restype = None
    L.oracle_align_association.argtypes = [pp, cp, cp, fp, fp, ip, C.c_int, C.POINTER(OracleAssociation)]
    L.oracle_align_association.restype = C.c_int
    return L


def _f(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def params_from(p):
    """Builds OracleParams from any object with the CvoParams attribute names (or a dict)."""
    get = (lambda k: p[k]) if isinstance(p, dict) else (lambda k: getattr(p, k))
    o = OracleParams()
    for name, _ in OracleParams._fields_:
        setattr(o, name, get(name))
    return o


class Cloud:
    """Keeps the numpy arrays alive next to the C struct."""

    def __init__(self, xyz, feat=None, label=None, geo=None):
        self.xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        n = self.xyz.shape[0]
        self.feat = None if feat is None else np.ascontiguousarray(feat, np.float32).reshape(n, FD)
        self.label = None if label is None else np.ascontiguousarray(label, np.float32).reshape(n, NC)
        self.geo = None if geo is None else np.ascontiguousarray(geo, np.float32).reshape(n, 2)
        self.c = OracleCloud(n, _f(self.xyz), _f(self.feat), _f(self.label), _f(self.geo))

    @property
    def n(self):
        return self.xyz.shape[0]

    @classmethod
    def from_pointcloud(cls, pc):
        """From a unified_cvo_amd.CvoPointCloud, through the same conversion the device upload uses."""
        xyz, feat, label, geo = pc.device_arrays()
        return cls(xyz, feat, label, geo)


def _cm(T):
    return np.ascontiguousarray(np.asarray(T, np.float32).reshape(4, 4).T).reshape(16)


def cubic_roots(coef):
    c = np.asarray(coef, np.float64)
    re, im = np.zeros(3), np.zeros(3)
    dp = C.POINTER(C.c_double)
    lib().oracle_cubic_roots(c.ctypes.data_as(dp), re.ctypes.data_as(dp), im.ctypes.data_as(dp))
    return re + 1j * im


def select_step(B, Cc, D, E, min_step, max_step):
    return lib().oracle_select_step(B, Cc, D, E, min_step, max_step)


def exp_sek3(xi, dt):
    x = np.asarray(xi, np.float32)
    out = np.zeros(12, np.float32)
    lib().oracle_exp_sek3(_f(x), dt, _f(out))
    return out.reshape(3, 4)


def se3_log_norm(dR, dT):
    r = np.ascontiguousarray(dR, np.float64).reshape(9)
    t = np.ascontiguousarray(dT, np.float64).reshape(3)
    dp = C.POINTER(C.c_double)
    return lib().oracle_se3_log_norm(r.ctypes.data_as(dp), t.ctypes.data_as(dp))


def indicator_run(indicators, window, thr):
    x = np.ascontiguousarray(indicators, np.float32)
    out = np.zeros(x.shape[0], np.uint8)
    lib().oracle_indicator_run(_f(x), x.shape[0], window, thr, out.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return out.astype(bool)


def transform_cloud(R, T, y0):
    """update_tf + transform_pointcloud: returns (Rinv, Tinv, yt)."""
    R = np.ascontiguousarray(R, np.float32).reshape(9)
    T = np.ascontiguousarray(T, np.float32).reshape(3)
    Ri, Ti = np.zeros(9, np.float32), np.zeros(3, np.float32)
    lib().oracle_update_tf(_f(R), _f(T), _f(Ri), _f(Ti))
    y0 = np.ascontiguousarray(y0, np.float32).reshape(-1, 3)
    yt = np.zeros_like(y0)
    lib().oracle_transform(_f(Ri), _f(Ti), y0.shape[0], _f(y0), _f(yt))
    return Ri.reshape(3, 3), Ti, yt


def transform_pose_vec(pose12, xyz):
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    out = np.zeros_like(xyz)
    pose = np.ascontiguousarray(np.asarray(pose12, np.float64).reshape(12).astype(np.float32))
    lib().oracle_transform_pose_vec(_f(pose), xyz.shape[0], _f(xyz), _f(out))
    return out


def se_kernel(p, x, y_transformed, K, ell, literal=False):
    n = x.n
    mat = np.zeros((n, K), np.float32)
    ind = np.zeros((n, K), np.int32)
    nz = np.zeros(n, np.uint32)
    lib().oracle_se_kernel(C.byref(p), C.byref(x.c), C.byref(y_transformed.c), K, ell, _f(mat),
                           ind.ctypes.data_as(C.POINTER(C.c_int)), nz.ctypes.data_as(C.POINTER(C.c_uint)),
                           1 if literal else 0)
    return mat, ind, nz


def iteration(p, x, y, R, T, ell, K, want_ell=False):
    """One loop body of align_impl on the given state.  Returns dict (R, T updated)."""
    R = np.ascontiguousarray(R, np.float32).reshape(9).copy()
    T = np.ascontiguousarray(T, np.float32).reshape(3).copy()
    tr = OracleTrace()
    ret = C.c_int(0)
    mat = ind = nz = None
    if want_ell:
        mat = np.zeros((x.n, K), np.float32)
        ind = np.zeros((x.n, K), np.int32)
        nz = np.zeros(x.n, np.uint32)
    status = lib().oracle_iteration(
        C.byref(p), C.byref(x.c), C.byref(y.c), _f(R), _f(T), ell, K, C.byref(tr), C.byref(ret), _f(mat),
        None if ind is None else ind.ctypes.data_as(C.POINTER(C.c_int)),
        None if nz is None else nz.ctypes.data_as(C.POINTER(C.c_uint)))
    return dict(status=status, ret=ret.value, trace=tr, R=R.reshape(3, 3), T=T, mat=mat, ind=ind, nonzeros=nz)


def align(p, x, y, init, trace_capacity=0, trace_dense=0, trace_every=0, max_iterations=0):
    out = np.zeros(16, np.float32)
    iters = C.c_int(0)
    nt = C.c_int(0)
    secs = C.c_double(0)
    tr = (OracleTrace * max(trace_capacity, 1))()
    init_c = _cm(init)
    ret = lib().oracle_align(C.byref(p), C.byref(x.c), C.byref(y.c), _f(init_c), _f(out), C.byref(iters),
                             tr if trace_capacity > 0 else None, trace_capacity, trace_dense, trace_every,
                             C.byref(nt), C.byref(secs), max_iterations)
    return dict(ret=ret, transform=out.reshape(4, 4).T.copy(), iterations=iters.value,
                trace=[tr[i] for i in range(nt.value)], seconds=secs.value)


def align_association(p, x, y, init, max_iterations=0):
    """align() with is_exporting_association (CvoGPU.cu:1552-1556): the pose plus the exported Association as
    (row, col, val) triplets in row order, the source inliers, and the two strides (written / read)."""
    out = np.zeros(16, np.float32)
    iters = C.c_int(0)
    cap = x.n * min(p.nearest_neighbors_max, max(y.n, 1)) + 1
    row, col = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    val = np.zeros(cap, np.float32)
    inl = np.zeros(x.n + 1, np.int32)
    ipt = C.POINTER(C.c_int)
    A = OracleAssociation(cap, x.n + 1, row.ctypes.data_as(ipt), col.ctypes.data_as(ipt), _f(val), inl.ctypes.data_as(ipt),
                          0, 0, 0, 0, 0)
    ret = lib().oracle_align_association(C.byref(p), C.byref(x.c), C.byref(y.c), _f(_cm(init)), _f(out), C.byref(iters),
                                         max_iterations, C.byref(A))
    assert not A.overflow
    n = A.n_pairs
    return dict(ret=ret, transform=out.reshape(4, 4).T.copy(), iterations=iters.value, row=row[:n].copy(),
                col=col[:n].copy(), val=val[:n].copy(), source_inliers=inl[:A.n_source_inliers].copy(),
                K_used=A.K_used, K_final=A.K_final)


def inner_product(p, x, y, T, ell):
    return lib().oracle_inner_product(C.byref(p), C.byref(x.c), C.byref(y.c), _f(_cm(T)), ell)


def function_angle(p, x, y, T, ell, is_approximate=True):
    return lib().oracle_function_angle(C.byref(p), C.byref(x.c), C.byref(y.c), _f(_cm(T)), ell,
                                       1 if is_approximate else 0)


def association(p, x, y, T, ell):
    n, K = x.n, p.nearest_neighbors_max
    row_ptr = np.zeros(n + 1, np.int32)
    col = np.zeros(n * min(K, y.n) + 1, np.int32)
    val = np.zeros(n * min(K, y.n) + 1, np.float32)
    ip = C.POINTER(C.c_int)
    cnt = lib().oracle_association(C.byref(p), C.byref(x.c), C.byref(y.c), _f(_cm(T)), ell,
                                   row_ptr.ctypes.data_as(ip), col.ctypes.data_as(ip), _f(val))
    return row_ptr, col[:cnt], val[:cnt]


def association_non_isotropic(p, x, y, T, kernel):
    """CSR (row_ptr, col, val) + the restated inverse of `kernel` (3x3)."""
    n = x.n
    K = p.nearest_neighbors_max
    cap = n * min(K, y.n)
    row_ptr = np.zeros(n + 1, np.int32)
    col = np.zeros(max(cap, 1), np.int32)
    val = np.zeros(max(cap, 1), np.float32)
    kcm = np.ascontiguousarray(np.asarray(kernel, np.float32).reshape(3, 3).T).reshape(9)
    kinv = np.zeros(9, np.float32)
    cnt = lib().oracle_association_non_isotropic(C.byref(p), C.byref(x.c), C.byref(y.c), _f(_cm(T)), _f(kcm),
                                                 row_ptr.ctypes.data_as(C.POINTER(C.c_int)),
                                                 col.ctypes.data_as(C.POINTER(C.c_int)), _f(val), _f(kinv))
    return row_ptr, col[:cnt], val[:cnt], kinv.reshape(3, 3)


def scan_seconds(reset=True):
    """Seconds the oracle spent in the association scan (se_kernel) since the last reset."""
    return float(lib().oracle_scan_seconds(1 if reset else 0))


def set_grid(on):
    lib().oracle_set_grid(1 if on else 0)


def num_threads():
    return lib().oracle_num_threads()


def set_num_threads(n):
    lib().oracle_set_num_threads(n)
