"""ctypes binding of the C-ABI in include/cvo_hip.h.

The library is the product: there is no Python / CPU fallback.  Importing this module when
libcvo_hip.so is missing raises, loudly.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libcvo_hip.so")

CVO_OK = 0
CVO_RET_FLOW_VANISHED = -1
CVO_E_INVALID = -2
CVO_E_HIP = -3
CVO_E_NOMEM = -4
CVO_E_UNSUPPORTED = -5
CVO_E_VERIFY = -6


class cvo_params_t(C.Structure):
    """Layout-identical to cvo::CvoParams (CvoParams.hpp:12-73) / cvo_params_t (cvo_hip.h)."""

    _fields_ = [
        ("ell_init_first_frame", C.c_float),
        ("ell_init", C.c_float),
        ("ell_min", C.c_float),
        ("min_ell_iter_limit", C.c_int),
        ("ell_max", C.c_float),
        ("dl", C.c_double),
        ("dl_step", C.c_double),
        ("sigma", C.c_float),
        ("sp_thres", C.c_float),
        ("c", C.c_float),
        ("d", C.c_float),
        ("c_ell", C.c_float),
        ("c_sigma", C.c_float),
        ("s_ell", C.c_float),
        ("s_sigma", C.c_float),
        ("MAX_ITER", C.c_int),
        ("eps", C.c_float),
        ("eps_2", C.c_float),
        ("min_step", C.c_float),
        ("max_step", C.c_float),
        ("step", C.c_float),
        ("nearest_neighbors_max", C.c_int),
        ("ell_decay_rate", C.c_float),
        ("ell_decay_rate_first_frame", C.c_float),
        ("ell_decay_start", C.c_int),
        ("ell_decay_start_first_frame", C.c_int),
        ("indicator_window_size", C.c_int),
        ("indicator_stable_threshold", C.c_float),
        ("is_pcl_visualization_on", C.c_int),
        ("is_using_least_square", C.c_int),
        ("is_ell_adaptive", C.c_int),
        ("is_full_ip_matrix", C.c_int),
        ("is_using_geometry", C.c_int),
        ("is_using_intensity", C.c_int),
        ("is_using_semantics", C.c_int),
        ("is_using_range_ell", C.c_int),
        ("is_using_kdtree", C.c_int),
        ("is_exporting_association", C.c_int),
        ("is_using_geometric_type", C.c_int),
        ("multiframe_using_cpu", C.c_int),
        ("multiframe_max_iters", C.c_int),
        ("multiframe_ell_init", C.c_float),
        ("multiframe_ell_min", C.c_float),
        ("multiframe_iter_per_ell", C.c_int),
        ("multiframe_ell_decay_rate", C.c_float),
        ("multiframe_iterations_per_ell", C.c_int),
        ("multiframe_iterations_per_solve", C.c_int),
        ("multiframe_expected_points", C.c_int),
        ("multiframe_downsample_voxel_size", C.c_float),
        ("multiframe_num_neighbors", C.c_int),
        ("multiframe_least_squares_num_threads", C.c_int),
        ("multiframe_min_nonzeros", C.c_int),
    ]


class cvo_trace_t(C.Structure):
    _fields_ = [
        ("k", C.c_int),
        ("K", C.c_int),
        ("ell", C.c_float),
        ("step", C.c_float),
        ("nnz", C.c_uint),
        ("max_nnz", C.c_uint),
        ("omega", C.c_float * 3),
        ("v", C.c_float * 3),
        ("B", C.c_double),
        ("C", C.c_double),
        ("D", C.c_double),
        ("E", C.c_double),
        ("dist", C.c_double),
        ("R", C.c_float * 9),
        ("T", C.c_float * 3),
    ]


class cvo_align_info_t(C.Structure):
    _fields_ = [
        ("iterations", C.c_int),
        ("ret", C.c_int),
        ("final_ell", C.c_float),
        ("final_num_neighbors", C.c_int),
        ("seconds", C.c_double),
    ]


class cvo_align_opts_t(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int),
        ("override_state", C.c_int),
        ("ell0", C.c_float),
        ("K0", C.c_int),
        ("trace", C.POINTER(cvo_trace_t)),
        ("trace_capacity", C.c_int),
        ("trace_dense", C.c_int),
        ("trace_every", C.c_int),
        ("n_trace", C.POINTER(C.c_int)),
        ("iters_per_launch", C.c_int),
        ("use_graph", C.c_int),
        ("kernel_clock", C.c_int),
    ]


class cvo_batch_result_t(C.Structure):
    _fields_ = [
        ("ticket", C.c_longlong),
        ("transform", C.c_float * 16),
        ("info", cvo_align_info_t),
    ]


# every symbol include/cvo_hip.h declares (tests/test_capi_symbols.py checks the two lists agree)
EXPORTED = [
    "cvo_params_default", "cvo_ctx_create", "cvo_ctx_destroy", "cvo_last_error", "cvo_ctx_stream",
    "cvo_ctx_synchronize", "cvo_cloud_upload", "cvo_cloud_upload_aos192", "cvo_cloud_size", "cvo_cloud_free",
    "cvo_align", "cvo_align_ex", "cvo_align_batch", "cvo_batch_poses_to_device", "cvo_inner_product",
    "cvo_function_angle", "cvo_association", "cvo_association_non_isotropic", "cvo_cloud_transformed", "cvo_edge_kernel_matrix", "cvo_debug_last_ell", "cvo_debug_time_scan", "cvo_debug_time_kernels",
    "cvo_debug_kernel_clock",
    "cvo_debug_last_candidates", "cvo_debug_list_builds", "cvo_debug_row_classes", "cvo_debug_scan_stats", "cvo_debug_last_geometry", "cvo_version",
    "cvo_align_association", "cvo_debug_scalar_math", "cvo_debug_verified_rows", "cvo_debug_device_memory", "cvo_cloud_upload_many",
    "cvo_ctx_set_option", "cvo_ctx_advice", "cvo_debug_cloud_order",
    "cvo_process_hint_hw_queues", "cvo_shutdown",
    "cvo_batch_open", "cvo_batch_submit", "cvo_batch_poll", "cvo_batch_pending", "cvo_batch_stats", "cvo_batch_close",
]

_libs = {}


def lib(path=None):
    """Loads libcvo_hip.so (once; `path`: another build of the same C-ABI, e.g. an experiment build of build.build_variant).  Raises if the
    HIP extension has not been built."""
    path = os.path.abspath(path) if path else LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -m unified_cvo_amd.build` "
            "(there is no CPU fallback for the hot path)")
    L = C.CDLL(path)
    # process-wide: GPU_MAX_HW_QUEUES=8 unless the caller chose a value (include/cvo_hip.h, hardware queues).  Here, right
    # after the load: before this process's first HIP call if the library is loaded before anything touches the GPU
    L.cvo_process_hint_hw_queues()
    vp, ip, fp = C.c_void_p, C.c_int, C.POINTER(C.c_float)
    L.cvo_version.restype = C.c_char_p
    L.cvo_params_default.argtypes = [C.POINTER(cvo_params_t)]
    L.cvo_params_default.restype = None
    L.cvo_ctx_create.argtypes = [ip, C.POINTER(vp)]
    L.cvo_ctx_destroy.argtypes = [vp]
    L.cvo_ctx_destroy.restype = None
    L.cvo_ctx_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.cvo_last_error.argtypes = [vp]
    L.cvo_last_error.restype = C.c_char_p
    L.cvo_ctx_advice.argtypes = [vp]
    L.cvo_ctx_advice.restype = C.c_char_p
    L.cvo_ctx_stream.argtypes = [vp]
    L.cvo_ctx_stream.restype = vp
    L.cvo_ctx_synchronize.argtypes = [vp]
    L.cvo_cloud_upload.argtypes = [vp, ip, fp, fp, fp, fp, C.POINTER(vp)]
    L.cvo_cloud_upload_aos192.argtypes = [vp, ip, vp, C.POINTER(vp)]
    L.cvo_cloud_upload_many.argtypes = [vp, ip, C.POINTER(C.c_int), C.POINTER(fp), C.POINTER(fp), C.POINTER(fp),
                                        C.POINTER(fp), ip, C.POINTER(vp)]
    L.cvo_cloud_size.argtypes = [vp]
    L.cvo_cloud_free.argtypes = [vp]
    L.cvo_cloud_free.restype = None
    L.cvo_align.argtypes = [vp, C.POINTER(cvo_params_t), vp, vp, fp, fp, C.POINTER(cvo_align_info_t)]
    L.cvo_align_ex.argtypes = [vp, C.POINTER(cvo_params_t), vp, vp, fp, fp, C.POINTER(cvo_align_info_t),
                               C.POINTER(cvo_align_opts_t)]
    L.cvo_align_batch.argtypes = [vp, C.POINTER(cvo_params_t), ip, C.POINTER(vp), C.POINTER(vp), fp, fp,
                                  C.POINTER(cvo_align_info_t), C.POINTER(cvo_align_opts_t)]
    L.cvo_batch_poses_to_device.argtypes = [vp, vp, ip]
    L.cvo_batch_open.argtypes = [vp, C.POINTER(cvo_params_t), ip, ip, ip, ip, C.POINTER(cvo_align_opts_t), C.POINTER(vp)]
    L.cvo_batch_submit.argtypes = [vp, vp, vp, fp, ip, C.POINTER(C.c_longlong)]
    L.cvo_batch_poll.argtypes = [vp, ip, ip, C.POINTER(cvo_batch_result_t), C.POINTER(C.c_int)]
    L.cvo_batch_pending.argtypes = [vp]
    L.cvo_batch_stats.argtypes = [vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    L.cvo_batch_close.argtypes = [vp]
    L.cvo_batch_close.restype = None
    L.cvo_inner_product.argtypes = [vp, C.POINTER(cvo_params_t), vp, vp, fp, C.c_float, fp]
    L.cvo_function_angle.argtypes = [vp, C.POINTER(cvo_params_t), vp, vp, fp, C.c_float, ip, fp]
    L.cvo_association.argtypes = [vp, C.POINTER(cvo_params_t), vp, vp, fp, C.c_float, C.POINTER(C.c_int),
                                  C.POINTER(C.c_int), fp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.cvo_association_non_isotropic.argtypes = [vp, C.POINTER(cvo_params_t), vp, vp, fp, fp, C.POINTER(C.c_int),
                                                 C.POINTER(C.c_int), fp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.cvo_cloud_transformed.argtypes = [vp, vp, fp, C.POINTER(vp)]
    L.cvo_edge_kernel_matrix.argtypes = [vp, C.POINTER(cvo_params_t), vp, vp, C.c_float, ip, fp, C.POINTER(C.c_int), C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
    L.cvo_debug_last_ell.argtypes = [vp, ip, fp, C.POINTER(C.c_int), C.POINTER(C.c_uint)]
    L.cvo_debug_time_scan.argtypes = [vp, ip, fp]
    L.cvo_debug_time_kernels.argtypes = [vp, ip, fp, fp]
    L.cvo_debug_kernel_clock.argtypes = [vp, fp, fp, C.POINTER(C.c_ulonglong)]
    L.cvo_debug_last_candidates.argtypes = [vp, C.POINTER(C.c_ulonglong)]
    L.cvo_debug_list_builds.argtypes = [vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    L.cvo_debug_row_classes.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.cvo_debug_last_geometry.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.cvo_debug_scan_stats.argtypes = [vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.cvo_align_association.argtypes = [vp, ip, C.POINTER(C.c_int), C.POINTER(C.c_int), fp, C.c_size_t,
                                        C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.cvo_debug_scalar_math.argtypes = [vp, ip, ip, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.cvo_debug_verified_rows.argtypes = [vp, C.POINTER(C.c_ulonglong)]
    L.cvo_debug_device_memory.argtypes = [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.cvo_debug_cloud_order.argtypes = [vp, C.POINTER(C.c_int)]
    for name in EXPORTED:
        getattr(L, name)  # AttributeError here = the library does not export what the header declares
    _libs[path] = L
    return L
