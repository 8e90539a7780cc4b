"""Python mirror of the reference's operator interface for the path (cvo::CvoGPU, cvo::CvoPointCloud;
include/UnifiedCvo/cvo/CvoGPU.hpp:49-229, utils/CvoPointCloud.hpp:126-188), over the C-ABI.

This is the harness used by tests/ and bench.py; the C++ veneer with the same names lives in
include/UnifiedCvo/.  All compute happens in libcvo_hip.so on the GPU.
"""
import ctypes as C
import weakref
import os

import numpy as np

from . import _capi
from .params import CvoParams, read_cvo_params_yaml
from .synth import FEATURE_DIMENSIONS, NUM_CLASSES


class CvoError(RuntimeError):
    pass


def _fptr(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


class CvoPointCloud:
    """Host container: positions (n,3), features (n,F), labels (n,C), geometric_types (n,2)."""

    def __init__(self, feature_dimensions=0, num_classes=0):
        self.num_points_ = 0
        self.feature_dimensions_ = feature_dimensions
        self.num_classes_ = num_classes
        self.positions_ = np.zeros((0, 3), np.float32)
        self.features_ = np.zeros((0, feature_dimensions), np.float32)
        self.labels_ = np.zeros((0, num_classes), np.float32)
        self.geometric_types_ = np.zeros((0, 2), np.float32)
        self._reserved = False

    # -- constructors mirroring the pcl ones (CvoPointCloud.cpp:569-652) -------------------------
    @classmethod
    def from_xyz(cls, xyz):
        """pcl::PointXYZ constructor: F = 0, geometric_type = (1, 0) (CvoPointCloud.cpp:633-652)."""
        pc = cls(0, 0)
        pc.positions_ = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        pc.num_points_ = pc.positions_.shape[0]
        pc.geometric_types_ = np.tile(np.array([[1.0, 0.0]], np.float32), (pc.num_points_, 1))
        pc._reserved = True
        return pc

    @classmethod
    def from_xyzrgb(cls, xyz, rgb_u8):
        """pcl::PointXYZRGB constructor: features (r,g,b)/255,0,0; type (0,1) (CvoPointCloud.cpp:569-594)."""
        pc = cls(5, 0)
        pc.positions_ = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        pc.num_points_ = pc.positions_.shape[0]
        f = np.zeros((pc.num_points_, 5), np.float32)
        f[:, :3] = (np.asarray(rgb_u8).astype(np.int32).astype(np.float32) / np.float32(255.0))
        pc.features_ = f
        pc.geometric_types_ = np.tile(np.array([[0.0, 1.0]], np.float32), (pc.num_points_, 1))
        pc._reserved = True
        return pc

    @classmethod
    def from_arrays(cls, xyz, features=None, labels=None, geometric_types=None):
        n = np.asarray(xyz).reshape(-1, 3).shape[0]
        F = 0 if features is None else np.asarray(features).shape[1]
        Cn = 0 if labels is None else np.asarray(labels).shape[1]
        pc = cls(F, Cn)
        pc.reserve(n, F, Cn)
        pc.positions_[:] = np.asarray(xyz, np.float32).reshape(-1, 3)
        if F:
            pc.features_[:] = features
        if Cn:
            pc.labels_[:] = labels
        if geometric_types is not None:
            pc.geometric_types_[:] = geometric_types
        return pc

    # -- reserve / add_point (CvoPointCloud.cpp:1384-1420) ----------------------------------------
    def reserve(self, num_points, feature_dims, num_classes):
        self.num_points_ = num_points
        self.feature_dimensions_ = feature_dims
        self.num_classes_ = num_classes
        self.positions_ = np.zeros((num_points, 3), np.float32)
        self.features_ = np.zeros((num_points, feature_dims), np.float32)
        self.labels_ = np.zeros((num_points, num_classes), np.float32)
        self.geometric_types_ = np.zeros((num_points, 2), np.float32)
        self._reserved = True

    def add_point(self, index, xyz, feature, label, geometric_type):
        if index >= self.num_points_:
            return -1
        if (not self._reserved or self.features_.shape[0] < self.num_points_
                or self.features_.shape[1] != self.feature_dimensions_ or len(geometric_type) != 2):
            return -1
        self.positions_[index] = xyz
        if self.feature_dimensions_:
            self.features_[index] = feature
        if self.num_classes_:
            self.labels_[index] = label
        self.geometric_types_[index] = geometric_type
        return 0

    # -- getters (CvoPointCloud.hpp:140-154) ------------------------------------------------------
    def num_points(self):
        return self.num_points_

    size = num_points

    def num_classes(self):
        return self.num_classes_

    def num_features(self):
        return self.feature_dimensions_

    feature_dimensions = num_features

    def positions(self):
        return self.positions_

    def features(self):
        return self.features_

    def labels(self):
        return self.labels_

    semantics = labels

    # label_at / feature_at / geometry_type_at (CvoPointCloud.hpp:141-143, CvoPointCloud.cpp:1282-1286): rows by value
    def label_at(self, index):
        return np.array(self.labels_[index], dtype=np.float32)

    def feature_at(self, index):
        return np.array(self.features_[index], dtype=np.float32)

    def geometry_type_at(self, index):
        return np.array(self.geometric_types_.reshape(-1)[2 * index:2 * index + 2], dtype=np.float32)

    def geometric_types(self):
        return self.geometric_types_.reshape(-1)

    @staticmethod
    def transform(pose, inp, out):
        """static CvoPointCloud::transform (CvoPointCloud.cpp:1366-1382); feature_dimensions_ is not copied."""
        P = np.asarray(pose, np.float32).reshape(4, 4)
        out.num_points_ = inp.num_points_
        out.num_classes_ = inp.num_classes_
        out.features_ = inp.features_.copy()
        out.labels_ = inp.labels_.copy()
        out.positions_ = (inp.positions_ @ P[:3, :3].T + P[:3, 3]).astype(np.float32)
        out.geometric_types_ = inp.geometric_types_.copy()
        out._reserved = True

    def __add__(self, other):
        """operator+ concatenation (CvoPointCloud.cpp:1139-1151)."""
        r = CvoPointCloud(self.feature_dimensions_, self.num_classes_)
        r.num_points_ = self.num_points_ + other.num_points_
        r.positions_ = np.concatenate([self.positions_, other.positions_])
        r.features_ = np.concatenate([self.features_, other.features_]) if self.feature_dimensions_ else self.features_
        r.labels_ = np.concatenate([self.labels_, other.labels_]) if self.num_classes_ else self.labels_
        r.geometric_types_ = np.concatenate([self.geometric_types_, other.geometric_types_])
        r._reserved = True
        return r

    # -- what CvoPointCloud_to_gpu builds per point (CvoGPU_impl.cu:206-263) ----------------------
    def device_arrays(self):
        n = self.num_points_
        xyz = np.ascontiguousarray(self.positions_, np.float32)
        feat = None
        if self.features_.shape[0] == n and self.features_.shape[1] > 0:
            feat = np.zeros((n, FEATURE_DIMENSIONS), np.float32)
            k = min(FEATURE_DIMENSIONS, self.features_.shape[1])
            feat[:, :k] = self.features_[:, :k]
        label = None
        if self.num_classes_ > 0:
            label = np.zeros((n, NUM_CLASSES), np.float32)
            k = min(NUM_CLASSES, self.labels_.shape[1])
            label[:, :k] = self.labels_[:, :k]
        geo = np.ascontiguousarray(self.geometric_types_, np.float32).reshape(n, 2)
        return xyz, feat, label, geo


class DeviceCloud:
    """A cloud resident in HBM (cvo_cloud*)."""

    def __init__(self, gpu, pc):
        self.gpu = gpu
        self.n = pc.num_points()
        xyz, feat, label, geo = pc.device_arrays()
        self._keep = (xyz, feat, label, geo)
        h = C.c_void_p()
        rc = gpu.L.cvo_cloud_upload(gpu.ctx, self.n, _fptr(xyz), _fptr(feat), _fptr(label), _fptr(geo), C.byref(h))
        gpu._check(rc)
        self.handle = h

    def free(self):
        if self.handle:
            self.gpu.L.cvo_cloud_free(self.handle)
            self.handle = None

    def debug_order(self):
        """The cloud's spatial (k-d) ordering: original index of the point at every sorted position."""
        out = np.zeros(max(self.n, 1), np.int32)
        self.gpu._check(self.gpu.L.cvo_debug_cloud_order(self.handle, out.ctypes.data_as(C.POINTER(C.c_int))))
        return out[:self.n]

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceCloudAoS(DeviceCloud):
    """A cloud uploaded from the reference's 192-byte AoS records (pcl::PointCloud<CvoPoint>::points, the wire format
    of pcl_PointCloud_to_gpu, CvoGPU_impl.cu:287-362) through cvo_cloud_upload_aos192."""

    def __init__(self, gpu, records):
        rec = np.ascontiguousarray(records)
        assert rec.dtype.itemsize == 192, "CvoPoint is 192 bytes (PointSegmentedDistribution.hpp:17-99)"
        self.gpu = gpu
        self.n = int(rec.shape[0])
        self._keep = rec
        h = C.c_void_p()
        gpu._check(gpu.L.cvo_cloud_upload_aos192(gpu.ctx, self.n, rec.ctypes.data_as(C.c_void_p), C.byref(h)))
        self.handle = h


# numpy view of pcl::PointSegmentedDistribution<5, 19> (PointSegmentedDistribution.hpp:17-99): byte offsets as laid out
# by PCL_ADD_POINT4D / PCL_ADD_RGB and the member order, verified with a layout-identical struct
CVO_POINT_DTYPE = np.dtype({
    "names": ["xyz", "pad_w", "rgba", "features", "label", "label_distribution", "geometric_type", "normal", "covariance",
              "cov_eigenvalues"],
    "formats": [(np.float32, 3), np.float32, np.uint32, (np.float32, 5), np.int32, (np.float32, 19), (np.float32, 2),
                (np.float32, 3), (np.float32, 9), (np.float32, 3)],
    "offsets": [0, 12, 16, 20, 40, 44, 120, 128, 140, 176],
    "itemsize": 192,
})


def cvo_points_from_pointcloud(pc):
    """What CvoPointCloud_to_gpu builds per point (CvoGPU_impl.cu:206-263) as an array of CvoPoint records: xyz,
    features (+ r, g, b bytes = min(255, f * 255)), label_distribution (+ label = argmax), geometric_type."""
    xyz, feat, label, geo = pc.device_arrays()
    n = xyz.shape[0]
    rec = np.zeros(n, CVO_POINT_DTYPE)
    rec["xyz"] = xyz
    rec["pad_w"] = 1.0
    if feat is not None:
        rec["features"] = feat
        rgb = np.minimum(255.0, feat[:, :3] * 255.0).astype(np.uint32)
        rec["rgba"] = (rgb[:, 0] << 16) | (rgb[:, 1] << 8) | rgb[:, 2]
    if label is not None:
        rec["label_distribution"] = label
        rec["label"] = np.argmax(label, axis=1)
    if geo is not None:
        rec["geometric_type"] = geo
    return rec


def _mat_to_c(T):
    a = np.ascontiguousarray(np.asarray(T, np.float32).reshape(4, 4).T).reshape(16)
    return a


class AlignResult:
    def __init__(self, ret, transform, info, trace=None):
        self.ret = ret
        self.transform = transform  # 4x4 float32 (row, col)
        self.iterations = info.iterations
        self.final_ell = info.final_ell
        self.final_num_neighbors = info.final_num_neighbors
        self.seconds = info.seconds
        self.trace = trace


class BatchQueue:
    """cvo_batch_open / _submit / _poll / _close: a stream of frame pairs through a fixed number of in-flight slots."""

    def __init__(self, gpu, slots, max_source_points, max_target_points, min_source_points=0, max_iterations=0):
        self.handle = None  # (set before anything can raise: close() / __del__ then have something to look at)
        self.gpu = gpu
        self.slots = slots
        o = _capi.cvo_align_opts_t()
        o.max_iterations = max_iterations
        p = gpu.params.to_ctypes()
        h = C.c_void_p()
        gpu._check(gpu.L.cvo_batch_open(gpu.ctx, C.byref(p), slots, max_source_points, max_target_points, min_source_points,
                                        C.byref(o), C.byref(h)))
        self.handle = h
        self._keep = {}
        gpu._queues.append(weakref.ref(self))  # (weak: a queue dropped without close() must still be collected)

    def submit(self, source, target, init, max_iterations=0):
        src, tgt = self.gpu._dev(source), self.gpu._dev(target)
        t = C.c_longlong()
        Tm = _mat_to_c(init)
        self.gpu._check(self.gpu.L.cvo_batch_submit(self.handle, src.handle, tgt.handle, _fptr(Tm), max_iterations, C.byref(t)))
        self._keep[t.value] = (src, tgt)  # the clouds must outlive the solve
        return t.value

    def poll(self, wait=1, capacity=None):
        """Finished pairs in submission order as AlignResult objects whose `.ticket` is the submission's ticket; wait:
        0 = just make progress, 1 = until one is ready, 2 = until everything submitted has finished."""
        cap = capacity or max(self.pending(), 1)
        buf = (_capi.cvo_batch_result_t * cap)()
        n = C.c_int()
        self.gpu._check(self.gpu.L.cvo_batch_poll(self.handle, wait, cap, buf, C.byref(n)))
        out = []
        for i in range(n.value):
            r = buf[i]
            T = np.array(list(r.transform), np.float32).reshape(4, 4).T.copy()
            res = AlignResult(r.info.ret, T, r.info)
            res.ticket = int(r.ticket)
            self._keep.pop(res.ticket, None)
            out.append(res)
        return out

    def pending(self):
        return int(self.gpu.L.cvo_batch_pending(self.handle))

    def stats(self):
        a, b, c = C.c_ulonglong(), C.c_ulonglong(), C.c_ulonglong()
        self.gpu._check(self.gpu.L.cvo_batch_stats(self.handle, C.byref(a), C.byref(b), C.byref(c)))
        return {"chunks": a.value, "full_chunks": b.value, "refills": c.value}

    def close(self):
        if self.handle:
            self.gpu.L.cvo_batch_close(self.handle)
            self.handle = None
            self._keep = {}
            self.gpu._queues[:] = [w for w in self.gpu._queues if w() is not None and w() is not self]

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CvoGPU:
    """cvo::CvoGPU(yaml) over the HIP backend."""

    def __init__(self, param_file=None, params=None, device=0, library=None):
        self.L = _capi.lib(library)  # (library: another build of the same C-ABI, e.g. an experiment build of build.build_variant)
        if params is not None:
            self.params = params
        elif param_file is not None:
            self.params = read_cvo_params_yaml(param_file)
        else:
            self.params = CvoParams()
        ctx = C.c_void_p()
        rc = self.L.cvo_ctx_create(device, C.byref(ctx))
        if rc != 0:
            raise CvoError(f"cvo_ctx_create(device={device}) failed with {rc}: is a HIP GPU visible?")
        self.ctx = ctx
        self._queues = []  # weak references to the live BatchQueue objects (closed before the context goes)

    def close(self):
        for w in list(getattr(self, "_queues", [])):
            q = w()
            if q is not None:
                q.close()
        if getattr(self, "ctx", None):
            self.L.cvo_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc <= _capi.CVO_E_INVALID:
            raise CvoError(f"error {rc}: {self.L.cvo_last_error(self.ctx).decode()}")
        return rc

    def get_params(self):
        return self.params

    def set_option(self, name, value):
        """cvo_ctx_set_option: a tuning / diagnostic switch of this context (the CVO_<NAME> environment variables are read
        once, when the context is created); value None clears it."""
        v = None if value is None else str(value).encode()
        self._check(self.L.cvo_ctx_set_option(self.ctx, name.encode(), v))

    def write_params(self, p):
        """CvoGPU::write_params (CvoGPU.cu:73-77): the device copy is refreshed per call here."""
        self.params = p

    def upload(self, pc):
        return DeviceCloud(self, pc)

    def upload_many(self, clouds, threads=None):
        """Uploads a list of clouds with cvo_cloud_upload_many: a pool of host threads inside the library, each cloud
        ordered / allocated / copied by one of them on its own stream (one ctypes call, the GIL is released for all of it)."""
        clouds = list(clouds)
        k = len(clouds)
        if k == 0:
            return []
        if threads is None:
            try:
                threads = len(os.sched_getaffinity(0))
            except AttributeError:
                threads = os.cpu_count() or 1
        threads = max(1, min(int(threads), k, 32))
        arrs = [pc.device_arrays() for pc in clouds]
        fpp = C.POINTER(C.c_float)
        n = (C.c_int * k)(*[a[0].shape[0] for a in arrs])

        def col(i):
            if all(a[i] is None for a in arrs):
                return None
            return (fpp * k)(*[_fptr(a[i]) if a[i] is not None else C.cast(None, fpp) for a in arrs])

        xyz = (fpp * k)(*[_fptr(a[0]) for a in arrs])
        out = (C.c_void_p * k)()
        self._check(self.L.cvo_cloud_upload_many(self.ctx, k, n, xyz, col(1), col(2), col(3), threads, out))
        res = []
        for i in range(k):
            d = DeviceCloud.__new__(DeviceCloud)
            d.gpu, d.n, d._keep, d.handle = self, int(n[i]), None, C.c_void_p(out[i])
            res.append(d)
        return res

    def upload_aos192(self, records):
        """pcl_PointCloud_to_gpu: uploads an array of 192-byte CvoPoint records (dtype CVO_POINT_DTYPE)."""
        return DeviceCloudAoS(self, records)

    def _dev(self, pc):
        return pc if isinstance(pc, DeviceCloud) else DeviceCloud(self, pc)

    def _opts(self, max_iterations=0, ell0=None, K0=None, trace_capacity=0, trace_dense=0, trace_every=0,
              n_pairs=1, iters_per_launch=0, use_graph=0, kernel_clock=False):
        o = _capi.cvo_align_opts_t()
        o.max_iterations = max_iterations
        keep = []
        if ell0 is not None or K0 is not None:
            o.override_state = 1
            o.ell0 = self.params.ell_init if ell0 is None else ell0
            o.K0 = self.params.nearest_neighbors_max if K0 is None else K0
        if trace_capacity > 0:
            tr = (_capi.cvo_trace_t * (trace_capacity * n_pairs))()
            nt = (C.c_int * n_pairs)()
            o.trace = tr
            o.trace_capacity = trace_capacity
            o.trace_dense = trace_dense
            o.trace_every = trace_every
            o.n_trace = nt
            keep = [tr, nt]
        o.iters_per_launch = iters_per_launch
        o.use_graph = use_graph
        o.kernel_clock = 1 if kernel_clock else 0
        return o, keep

    def align(self, source, target, T_target_frame_to_source_frame, **kw):
        """CvoGPU::align (CvoGPU.cu:1605-1632).  Returns AlignResult (ret = 0 / -1)."""
        if source.num_points() == 0 or target.num_points() == 0 if not isinstance(source, DeviceCloud) else False:
            info = _capi.cvo_align_info_t()
            return AlignResult(0, None, info)  # transform untouched
        src, tgt = self._dev(source), self._dev(target)
        p = self.params.to_ctypes()
        init = _mat_to_c(T_target_frame_to_source_frame)
        out = np.zeros(16, np.float32)
        info = _capi.cvo_align_info_t()
        opts, keep = self._opts(**kw)
        rc = self.L.cvo_align_ex(self.ctx, C.byref(p), src.handle, tgt.handle, _fptr(init), _fptr(out),
                                 C.byref(info), C.byref(opts))
        self._check(rc)
        trace = None
        if keep:
            trace = [keep[0][i] for i in range(keep[1][0])]
        self._keepalive = keep
        return AlignResult(rc, out.reshape(4, 4).T.copy(), info, trace)

    def align_batch(self, sources, targets, inits, **kw):
        """n independent pairs solved concurrently on this context's GPU (cvo_align_batch)."""
        n = len(sources)
        src = [self._dev(s) for s in sources]
        tgt = [self._dev(t) for t in targets]
        sh = (C.c_void_p * n)(*[s.handle for s in src])
        th = (C.c_void_p * n)(*[t.handle for t in tgt])
        init = np.concatenate([_mat_to_c(T) for T in inits]).astype(np.float32)
        out = np.zeros(16 * n, np.float32)
        infos = (_capi.cvo_align_info_t * n)()
        p = self.params.to_ctypes()
        opts, keep = self._opts(n_pairs=n, **kw)
        rc = self.L.cvo_align_batch(self.ctx, C.byref(p), n, sh, th, _fptr(init), _fptr(out), infos, C.byref(opts))
        self._check(rc)
        res = []
        cap = opts.trace_capacity
        for i in range(n):
            trace = None
            if keep:
                trace = [keep[0][i * cap + t] for t in range(keep[1][i])]
            res.append(AlignResult(infos[i].ret, out[16 * i:16 * i + 16].reshape(4, 4).T.copy(), infos[i], trace))
        self._keepalive = keep
        return res

    def open_queue(self, slots, max_source_points, max_target_points, min_source_points=0, max_iterations=0):
        """A batch queue (cvo_batch_open): `slots` pairs in flight, finished pairs hand their slot to the next submitted
        one at a chunk boundary, results in submission order."""
        return BatchQueue(self, slots, max_source_points, max_target_points, min_source_points, max_iterations)

    def align_stream(self, sources, targets, inits, slots=64, max_iterations=0, limits=None):
        """All pairs through a batch queue of `slots` in-flight slots; returns their AlignResults in submission order
        (limits: per-pair iteration limits)."""
        src = [self._dev(s) for s in sources]
        tgt = [self._dev(t) for t in targets]
        if not src:  # (the C++ veneer returns an empty result likewise)
            return []
        q = self.open_queue(slots, max(s.n for s in src), max(t.n for t in tgt), min(s.n for s in src), max_iterations)
        try:
            for k, (s, t, T) in enumerate(zip(src, tgt, inits)):
                q.submit(s, t, T, limits[k] if limits else 0)
            out = []
            while q.pending():
                out.extend(q.poll(wait=2))
            return out
        finally:
            q.close()

    def align_association(self, n_source, pair=0, capacity=None):
        """The Association `align()` exports under is_exporting_association (CvoGPU.cu:1552-1556): CSR of the kernel
        matrix of the last executed iteration of pair `pair` of the last align call, + (stride_written, stride_read)."""
        row_ptr = np.zeros(n_source + 1, np.int32)
        nnz, kw, kr = C.c_size_t(), C.c_int(), C.c_int()
        ipt = C.POINTER(C.c_int)
        rc = self.L.cvo_align_association(self.ctx, pair, row_ptr.ctypes.data_as(ipt), None, None, 0, C.byref(nnz),
                                          C.byref(kw), C.byref(kr))
        if rc != _capi.CVO_E_NOMEM:
            self._check(rc)
        cap = nnz.value if capacity is None else capacity
        col, val = np.zeros(max(cap, 1), np.int32), np.zeros(max(cap, 1), np.float32)
        self._check(self.L.cvo_align_association(self.ctx, pair, row_ptr.ctypes.data_as(ipt), col.ctypes.data_as(ipt),
                                                 _fptr(val), cap, C.byref(nnz), C.byref(kw), C.byref(kr)))
        return row_ptr, col[:nnz.value], val[:nnz.value], kw.value, kr.value

    def poses_to_device(self, dst_ptr, n):
        self._check(self.L.cvo_batch_poses_to_device(self.ctx, C.c_void_p(dst_ptr), n))

    def inner_product_gpu(self, source, target, T, ell):
        src, tgt = self._dev(source), self._dev(target)
        p = self.params.to_ctypes()
        out = C.c_float()
        Tm = _mat_to_c(T)
        self._check(self.L.cvo_inner_product(self.ctx, C.byref(p), src.handle, tgt.handle, _fptr(Tm), ell,
                                             C.byref(out)))
        return out.value

    def function_angle(self, source, target, T, ell, is_approximate=True):
        src, tgt = self._dev(source), self._dev(target)
        p = self.params.to_ctypes()
        out = C.c_float()
        Tm = _mat_to_c(T)
        self._check(self.L.cvo_function_angle(self.ctx, C.byref(p), src.handle, tgt.handle, _fptr(Tm), ell,
                                              1 if is_approximate else 0, C.byref(out)))
        return out.value

    def compute_association_gpu(self, source, target, T, lengthscale):
        """Association::pairs as CSR (row_ptr, col, val) (CvoGPU.cu:1876-1911)."""
        src, tgt = self._dev(source), self._dev(target)
        n = src.n
        p = self.params.to_ctypes()
        cap = n * min(self.params.nearest_neighbors_max, tgt.n)
        row_ptr = np.zeros(n + 1, np.int32)
        col = np.zeros(max(cap, 1), np.int32)
        val = np.zeros(max(cap, 1), np.float32)
        nnz = C.c_size_t()
        Tm = _mat_to_c(T)
        self._check(self.L.cvo_association(self.ctx, C.byref(p), src.handle, tgt.handle, _fptr(Tm), lengthscale,
                                           row_ptr.ctypes.data_as(C.POINTER(C.c_int)),
                                           col.ctypes.data_as(C.POINTER(C.c_int)), _fptr(val), cap, C.byref(nnz)))
        return row_ptr, col[:nnz.value], val[:nnz.value]

    def compute_association_gpu_non_isotropic(self, source, target, T, kernel):
        """Association under the Mahalanobis kernel d^T kernel^-1 d, as CSR (CvoGPU.cu:1913-1995)."""
        src, tgt = self._dev(source), self._dev(target)
        n = src.n
        p = self.params.to_ctypes()
        cap = n * min(self.params.nearest_neighbors_max, tgt.n)
        row_ptr = np.zeros(n + 1, np.int32)
        col = np.zeros(max(cap, 1), np.int32)
        val = np.zeros(max(cap, 1), np.float32)
        nnz = C.c_size_t()
        Tm = _mat_to_c(T)
        kcm = np.ascontiguousarray(np.asarray(kernel, np.float32).reshape(3, 3).T).reshape(9)
        self._check(self.L.cvo_association_non_isotropic(self.ctx, C.byref(p), src.handle, tgt.handle, _fptr(Tm), _fptr(kcm),
                                                         row_ptr.ctypes.data_as(C.POINTER(C.c_int)),
                                                         col.ctypes.data_as(C.POINTER(C.c_int)), _fptr(val), cap,
                                                         C.byref(nnz)))
        return row_ptr, col[:nnz.value], val[:nnz.value]

    # -- multi-frame edge kernel (BinaryStateGPU::update_inner_product, IRLS_State_GPU.cu:43-79) ----
    def transformed(self, cloud, pose_3x4):
        """CvoFrameGPU::transform_pointcloud: a new resident cloud moved by a 3x4 row-major pose."""
        src = self._dev(cloud)
        pose = np.ascontiguousarray(np.asarray(pose_3x4, np.float64).reshape(12).astype(np.float32))
        h = C.c_void_p()
        self._check(self.L.cvo_cloud_transformed(self.ctx, src.handle, _fptr(pose), C.byref(h)))
        out = DeviceCloud.__new__(DeviceCloud)
        out.gpu, out.n, out._keep, out.handle = self, src.n, None, h
        return out

    def edge_kernel_matrix(self, frame1, frame2, ell, num_neighbors):
        """fill_in_A_mat_gpu on two transformed frames -> (mat, ind, nonzeros, nonzero_sum) in the reference's host
        layout ([n1 x K] row-major, 0 / -1 padded)."""
        f1, f2 = self._dev(frame1), self._dev(frame2)
        K = int(num_neighbors)
        mat = np.zeros((f1.n, K), np.float32)
        ind = np.zeros((f1.n, K), np.int32)
        nz = np.zeros(f1.n, np.uint32)
        total = C.c_uint()
        p = self.params.to_ctypes()
        self._check(self.L.cvo_edge_kernel_matrix(self.ctx, C.byref(p), f1.handle, f2.handle, float(ell), K, _fptr(mat),
                                                  ind.ctypes.data_as(C.POINTER(C.c_int)),
                                                  nz.ctypes.data_as(C.POINTER(C.c_uint)), C.byref(total)))
        return mat, ind, nz, total.value

    # -- test / profiling hooks --------------------------------------------------------------------
    def debug_last_ell(self, n_rows, K):
        mat = np.zeros((n_rows, K), np.float32)
        ind = np.zeros((n_rows, K), np.int32)
        nz = np.zeros(n_rows, np.uint32)
        self._check(self.L.cvo_debug_last_ell(self.ctx, K, _fptr(mat), ind.ctypes.data_as(C.POINTER(C.c_int)),
                                              nz.ctypes.data_as(C.POINTER(C.c_uint))))
        return mat, ind, nz

    def debug_time_scan(self, reps=20):
        ms = C.c_float()
        self._check(self.L.cvo_debug_time_scan(self.ctx, reps, C.byref(ms)))
        return ms.value

    def debug_time_kernels(self, reps=20):
        a, c = C.c_float(), C.c_float()
        self._check(self.L.cvo_debug_time_kernels(self.ctx, reps, C.byref(a), C.byref(c)))
        return a.value, c.value

    def debug_kernel_clock(self):
        """(k_assoc ms, k_coeff ms, intervals): average duration per pair and launch inside the last align call's
        optimiser loop, from the device clock (needs CVO_KERNEL_CLOCK=1 in the environment at context creation)."""
        a, c, n = C.c_float(), C.c_float(), C.c_ulonglong()
        self._check(self.L.cvo_debug_kernel_clock(self.ctx, C.byref(a), C.byref(c), C.byref(n)))
        return a.value, c.value, n.value

    def debug_last_geometry(self):
        """(sub-batches of the last call, pairs per sub-batch): the k_scan launches a profiler sees."""
        g, p = C.c_int(), C.c_int()
        self._check(self.L.cvo_debug_last_geometry(self.ctx, C.byref(g), C.byref(p)))
        return g.value, p.value

    def debug_scan_stats(self):
        """(tiles executed by k_scan during the last align call, rows per tile, targets per tile)."""
        t, r, c = C.c_ulonglong(), C.c_int(), C.c_int()
        self._check(self.L.cvo_debug_scan_stats(self.ctx, C.byref(t), C.byref(r), C.byref(c)))
        return t.value, r.value, c.value

    def debug_list_builds(self):
        b, it, ce = C.c_ulonglong(), C.c_ulonglong(), C.c_ulonglong()
        self._check(self.L.cvo_debug_list_builds(self.ctx, C.byref(b), C.byref(it), C.byref(ce)))
        return b.value, it.value, ce.value

    def debug_row_classes(self, pair=0):
        """(overflow rows, rows scanned literally, dense regime) of pair `pair` as its last list build left them."""
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self._check(self.L.cvo_debug_row_classes(self.ctx, pair, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, bool(c.value)

    def advice(self):
        """Performance-relevant observations about the process set-up (cvo_ctx_advice): "" when there is nothing to say,
        e.g. a text about GPU_MAX_HW_QUEUES when HIP was initialised with fewer than 8 hardware queues."""
        return self.L.cvo_ctx_advice(self.ctx).decode()

    def debug_scalar_math(self, op, items):
        """Runs one of the device's scalar routines (k_scalar_math ops 0-6, 8-11) on `items` (n x <=16 doubles); returns
        n x 16 doubles.  op 7 (indicator windows): items = [window, threshold, x_0, ...], returns the n decisions.
        ops 8-11 (hoisted division / exp against the plain forms): eight operands per item, out[:, 2l] plain,
        out[:, 2l+1] hoisted."""
        dp = C.POINTER(C.c_double)
        if op == 7:
            a = np.ascontiguousarray(items, np.float64).reshape(-1)
            n = a.shape[0] - 2
            out = np.zeros(n, np.float64)
        else:
            it = np.atleast_2d(np.asarray(items, np.float64))
            n = it.shape[0]
            a = np.zeros((n, 16), np.float64)
            a[:, :it.shape[1]] = it
            out = np.zeros((n, 16), np.float64)
        self._check(self.L.cvo_debug_scalar_math(self.ctx, op, n, a.ctypes.data_as(dp), out.ctypes.data_as(dp)))
        return out

    def debug_device_memory(self):
        """(free, total) bytes of this context's device."""
        f, t = C.c_size_t(), C.c_size_t()
        self._check(self.L.cvo_debug_device_memory(self.ctx, C.byref(f), C.byref(t)))
        return f.value, t.value

    def debug_verified_rows(self):
        v = C.c_ulonglong()
        self._check(self.L.cvo_debug_verified_rows(self.ctx, C.byref(v)))
        return v.value

    def debug_last_candidates(self):
        v = C.c_ulonglong()
        self._check(self.L.cvo_debug_last_candidates(self.ctx, C.byref(v)))
        return v.value


class CvoFrameGPU:
    """cvo::CvoFrameGPU (CvoFrameGPU.hpp:14-36): a point cloud under a 3x4 row-major pose; transform_pointcloud()
    refreshes the transformed copy resident on the device."""

    def __init__(self, gpu, pts, poses):
        self.gpu = gpu
        self.points = pts
        self.pose_vec = np.asarray(poses, np.float64).reshape(12).copy()
        self._init = gpu.upload(pts)
        self._transformed = None
        self.transform_pointcloud()

    def transform_pointcloud(self):
        if self._transformed is not None:
            self._transformed.free()
        self._transformed = self.gpu.transformed(self._init, self.pose_vec)

    def points_transformed_gpu(self):
        return self._transformed


class BinaryStateGPU:
    """cvo::BinaryStateGPU (IRLS_State_GPU.hpp:21-89) without the Ceres half: update_inner_product() recomputes the
    edge's kernel matrix from the two frames' current transformed clouds, with the reference's neighbour-count
    adaptation (IRLS_State_GPU.cu:45-47)."""

    def __init__(self, frame1, frame2, num_neighbor, init_ell):
        self.frame1, self.frame2 = frame1, frame2
        self.init_num_neighbors = int(num_neighbor)
        self.num_neighbors = int(num_neighbor)
        self.ell = float(init_ell)
        self.iter = 0
        self.mat = self.ind = self.nonzeros = None
        self.nonzero_sum = 0

    def update_inner_product(self):
        last = int(self.nonzeros.max()) if self.nonzeros is not None and self.nonzeros.size else 0
        if last > 0:
            self.num_neighbors = min(self.init_num_neighbors, int(last * 1.1))
        gpu = self.frame1.gpu
        self.mat, self.ind, self.nonzeros, self.nonzero_sum = gpu.edge_kernel_matrix(
            self.frame1.points_transformed_gpu(), self.frame2.points_transformed_gpu(), self.ell, self.num_neighbors)
        self.iter += 1
        return self.nonzero_sum
