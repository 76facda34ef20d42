"""ctypes binding of libecal.so (include/ecal.h).  No torch objects cross this layer: device pointers are
passed as integers (``tensor.data_ptr()``), streams as ``torch.cuda.current_stream().cuda_stream``."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# every symbol include/ecal.h declares (tests check that the built library exports them all)
EXPORTED_SYMBOLS = [
    "ecal_abi_version", "ecal_init", "ecal_destroy", "ecal_strerror", "ecal_last_error", "ecal_sync",
    "ecal_dbscan_batch", "ecal_dbscan_batch_dev",
    "ecal_window_bounds_dev", "ecal_check_sorted_dev", "ecal_sort_events_dev", "ecal_slice_events_dev",
    "ecal_set_point_order", "ecal_get_point_order", "ecal_ref_bucket_step", "ecal_ref_pixel_hash",
    "ecal_comm_unique_id", "ecal_comm_init", "ecal_comm_destroy", "ecal_comm_size", "ecal_comm_rank", "ecal_comm_allreduce_sum_dev", "ecal_comm_allreduce",
    "ecal_circle_radius_threshold", "ecal_extract_batch_dev", "ecal_extract_batch_ordered_dev", "ecal_extract_batch_exact_dev", "ecal_cluster_order_list_dev", "ecal_set_median_ties", "ecal_set_tail_mode", "ecal_get_tail_mode", "ecal_calibrate_fisheye_views", "ecal_set_profile_ranges",
    "ecal_get_median_ties", "ecal_detect_fused_dev", "ecal_cluster_order_dev", "ecal_cluster_order",
    "ecal_stream_create", "ecal_stream_destroy", "ecal_stream_size", "ecal_stream_data", "ecal_detect_batch", "ecal_copy_dev",
    "ecal_grid_order_dev", "ecal_grid_order", "ecal_associate_dev", "ecal_associate", "ecal_pin_host", "ecal_unpin_host",
    "ecal_detect_stream_tiled", "ecal_gather_features_dev", "ecal_detect_pass", "ecal_detect_keyframes", "ecal_detect_keyframes_sharded", "ecal_detect_keyframes_cap_hint", "ecal_detect_keyframes_cap_hint_dev", "ecal_stream_create_from_file", "ecal_stream_times", "ecal_rectify_batch_dev", "ecal_rectify_batch",
    "ecal_solver_create", "ecal_solver_destroy", "ecal_solver_param_size", "ecal_solver_normal_size",
    "ecal_solver_num_chunks", "ecal_solver_evaluate_dev", "ecal_solver_evaluate", "ecal_residuals_dev", "ecal_residuals", "ecal_lm_default_options",
    "ecal_solver_solve", "ecal_inverse_radial_distortion", "ecal_solver_create_dev", "ecal_solver_num_residuals",
    "ecal_associate_ranges_dev", "ecal_ref_nth_element_f64", "ecal_solver_create_from_stream", "ecal_rectify_keyframes", "ecal_solver_time_shard_cuts",
    "ecal_slice_events_packed_dev", "ecal_dbscan_batch_packed_dev", "ecal_extract_batch_packed_dev", "ecal_unpack_points_dev",
    "ecal_calib_default_options", "ecal_calib_view_blocks_dev", "ecal_pnp_batch_dev", "ecal_pnp_batch", "ecal_pose_gates", "ecal_calibrate_views", "ecal_spline_fit", "ecal_spline_eval", "ecal_spline_so3_refine",
]


class PackedPoints(ctypes.Structure):
    """ecal_packed_points: d_xy16 [cap_points] u32 (x | y << 16), d_seg_fmt [2S] u32 (0 doubles, 1 packed, 3 both)."""
    _fields_ = [("d_xy16", ctypes.c_void_p), ("d_seg_fmt", ctypes.c_void_p)]


class RectifyParams(ctypes.Structure):
    """ecal_rectify_params (include/ecal.h)."""
    _fields_ = [("fx", ctypes.c_double), ("fy", ctypes.c_double), ("cx", ctypes.c_double), ("cy", ctypes.c_double),
                ("dist", ctypes.c_double * 5), ("width", ctypes.c_double), ("height", ctypes.c_double),
                ("rows", ctypes.c_uint32), ("cols", ctypes.c_uint32), ("asymmetric", ctypes.c_int),
                ("circle_radius", ctypes.c_double), ("fit_circle", ctypes.c_int), ("model", ctypes.c_int)]


import weakref

_live_contexts = weakref.WeakSet()


def sync_env():
    """Tests: the library reads its debug switches (ECAL_* environment variables) once per context, at ecal_init; after changing
    one under a live context, call this — every live Context re-reads them (ecal_debug_reload_env)."""
    for c in list(_live_contexts):
        c.reload_env()


class EcalError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("ecal status %d: %s" % (status, msg))
        self.status = status


def lib_path():
    return os.path.join(_HERE, "libecal.so")


def load_library():
    """Load libecal.so; raises (never falls back) if the HIP library has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    p = lib_path()
    if not os.path.exists(p):
        raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(or `make -C eventcalib_amd/csrc`). There is no CPU fallback." % p)
    # torch wheels bundle their own libamdhip64 under the same soname: whichever HIP runtime is mapped first serves
    # the whole process, and torch cannot find a GPU through the system one.  Let torch (when present) map its
    # runtime first; libecal.so then binds to that already-loaded library.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(p)
    vp, u32, i32, f64 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_double
    L.ecal_abi_version.restype = i32
    L.ecal_init.argtypes = [i32, ctypes.POINTER(vp)]
    L.ecal_init.restype = i32
    L.ecal_destroy.argtypes = [vp]
    L.ecal_destroy.restype = None
    L.ecal_strerror.argtypes = [i32]
    L.ecal_strerror.restype = ctypes.c_char_p
    L.ecal_last_error.argtypes = [vp]
    L.ecal_last_error.restype = ctypes.c_char_p
    L.ecal_sync.argtypes = [vp]
    L.ecal_sync.restype = i32
    L.ecal_dbscan_batch.argtypes = [vp, vp, vp, u32, f64, u32, vp, vp]
    L.ecal_dbscan_batch.restype = i32
    L.ecal_dbscan_batch_dev.argtypes = [vp, vp, vp, vp, u32, u32, u32, f64, u32, vp, vp, vp]
    L.ecal_dbscan_batch_dev.restype = i32
    u64 = ctypes.c_uint64
    L.ecal_comm_unique_id.argtypes = [vp]
    L.ecal_comm_unique_id.restype = i32
    L.ecal_comm_init.argtypes = [vp, vp, i32, i32]
    L.ecal_comm_init.restype = i32
    L.ecal_comm_destroy.argtypes = [vp]
    L.ecal_comm_destroy.restype = i32
    L.ecal_comm_size.argtypes = [vp]
    L.ecal_comm_size.restype = i32
    L.ecal_comm_rank.argtypes = [vp]
    L.ecal_comm_rank.restype = i32
    L.ecal_comm_allreduce_sum_dev.argtypes = [vp, vp, ctypes.c_size_t, vp]
    L.ecal_comm_allreduce_sum_dev.restype = i32
    L.ecal_set_point_order.argtypes = [vp, i32]
    L.ecal_set_point_order.restype = i32
    L.ecal_get_point_order.argtypes = [vp]
    L.ecal_get_point_order.restype = i32
    L.ecal_ref_bucket_step.argtypes = [i32]
    L.ecal_ref_bucket_step.restype = u64
    L.ecal_ref_pixel_hash.argtypes = [f64, f64]
    L.ecal_ref_pixel_hash.restype = u64
    L.ecal_window_bounds_dev.argtypes = [vp, vp, u64, vp, vp, u32, vp, vp, vp, vp]
    L.ecal_window_bounds_dev.restype = i32
    L.ecal_check_sorted_dev.argtypes = [vp, vp, u64, vp, vp]
    L.ecal_check_sorted_dev.restype = i32
    L.ecal_slice_events_dev.argtypes = [vp, vp, u64, vp, vp, vp, u32, u32, u32, vp, vp, vp, vp, vp, vp]
    L.ecal_slice_events_dev.restype = i32
    L.ecal_circle_radius_threshold.argtypes = [f64, f64, i32, i32, i32, f64, f64]
    L.ecal_circle_radius_threshold.restype = f64
    L.ecal_extract_batch_dev.argtypes = [vp, vp, vp, vp, vp, vp, u32, u32, u32, u32, f64, i32, u32, vp, vp, vp, vp, vp, vp]
    L.ecal_extract_batch_dev.restype = i32
    L.ecal_detect_fused_dev.argtypes = [vp, vp, ctypes.c_uint64, vp, vp, vp, u32, u32, u32, u32, f64, u32, u32, u32, f64, i32, u32,
                                        vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.ecal_detect_fused_dev.restype = i32
    L.ecal_grid_order_dev.argtypes = [vp, vp, vp, vp, u32, u32, u32, vp, vp, vp]
    L.ecal_grid_order_dev.restype = i32
    L.ecal_rectify_batch_dev.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, u32, vp, ctypes.POINTER(RectifyParams),
                                         vp, vp, vp, vp]
    L.ecal_rectify_batch_dev.restype = i32
    L.ecal_associate_dev.argtypes = [vp, vp, ctypes.c_uint64, vp, vp, u32, u32, f64, f64, f64, f64, vp, vp, vp, vp, vp]
    L.ecal_associate_dev.restype = i32
    L.ecal_associate_ranges_dev.argtypes = [vp, vp, ctypes.c_uint64, vp, vp, u32, u32, vp, u32, f64, f64, vp, vp, vp, vp, vp, vp]
    L.ecal_associate_ranges_dev.restype = i32
    L.ecal_copy_dev.argtypes = [vp, vp, vp, ctypes.c_size_t, vp, i32]
    L.ecal_copy_dev.restype = i32
    _LIB = L
    return L


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


class Context:
    """One ecal_ctx (one per host thread, like one DBSCAN instance per reference worker)."""

    def __init__(self, device=0):
        self._L = load_library()
        h = ctypes.c_void_p()
        st = self._L.ecal_init(int(device), ctypes.byref(h))
        if st != 0:
            raise EcalError(st, self._L.ecal_strerror(st).decode())
        self._h = h
        self.device = int(device)
        _live_contexts.add(self)

    def reload_env(self):
        """ecal_debug_reload_env: read the ECAL_* debug switches again (they are read once, at ecal_init)."""
        if getattr(self, "_h", None):
            self._L.ecal_debug_reload_env.argtypes = [ctypes.c_void_p]
            self._check(self._L.ecal_debug_reload_env(self._h))

    def close(self):
        self._calibrate_pipe = None   # (calibrate.calibrate_stream keeps its detection pipeline's arrays with the context)
        if getattr(self, "_h", None):
            _live_contexts.discard(self)
            self._L.ecal_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, st):
        if st != 0:
            msg = self._L.ecal_strerror(st).decode()
            extra = self._L.ecal_last_error(self._h).decode()
            raise EcalError(st, msg + (": " + extra if extra else ""))

    def sync(self):
        self._check(self._L.ecal_sync(self._h))

    # ---- DBSCAN (host buffers) ----
    def dbscan_batch(self, xy, slice_off, eps, minpts):
        """DBSCAN::Run per slice.  xy [N,2] float64, slice_off [S+1] uint32 -> (labels int32 [N], n_clusters uint32 [S])."""
        xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
        slice_off = np.ascontiguousarray(slice_off, dtype=np.uint32)
        S = slice_off.shape[0] - 1
        if S < 0:
            raise ValueError("slice_off needs at least one entry")
        labels = np.full(xy.shape[0], -1, dtype=np.int32)
        ncl = np.zeros(max(S, 0), dtype=np.uint32)
        self._check(self._L.ecal_dbscan_batch(self._h, _ptr(xy), _ptr(slice_off), S, float(eps), int(minpts),
                                              _ptr(labels), _ptr(ncl)))
        return labels, ncl

    # ---- DBSCAN (device buffers, raw pointers) ----
    def dbscan_batch_dev(self, d_xy, d_seg_off, d_seg_cnt, S, n_points, max_seg_points, eps, minpts, d_labels,
                         d_n_clusters, stream=0):
        self._check(self._L.ecal_dbscan_batch_dev(self._h, d_xy, d_seg_off, d_seg_cnt, int(S), int(n_points),
                                                  int(max_seg_points), float(eps), int(minpts), d_labels,
                                                  d_n_clusters, stream))

    # ---- in-library RCCL communicator (one rank per GPU) ----
    @staticmethod
    def comm_unique_id():
        """Rank 0: the 128 bytes every rank passes to comm_init (hand them over with any transport)."""
        buf = ctypes.create_string_buffer(128)
        rc = load_library().ecal_comm_unique_id(buf)
        if rc:
            raise EcalError(rc, "ecal_comm_unique_id")
        return buf.raw

    def comm_init(self, unique_id, rank, world_size):
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        self._check(self._L.ecal_comm_init(self._h, buf, int(rank), int(world_size)))

    def comm_destroy(self):
        self._check(self._L.ecal_comm_destroy(self._h))

    def comm_size(self):
        return self._L.ecal_comm_size(self._h)

    def comm_rank(self):
        return self._L.ecal_comm_rank(self._h)

    def comm_allreduce_fn(self):
        """(ALLREDUCE_FN, user pointer) that make a solve / calibration all-reduce through this context's communicator
        (options.allreduce = ecal_comm_allreduce, options.allreduce_user = the context) — collectives are explicit."""
        fn = ctypes.cast(self._L.ecal_comm_allreduce, ALLREDUCE_FN)
        return fn, ctypes.cast(self._h, ctypes.c_void_p)

    def comm_allreduce_sum_dev(self, d_buf, n_doubles, stream=0):
        self._check(self._L.ecal_comm_allreduce_sum_dev(self._h, d_buf, int(n_doubles), stream))

    # ---- ingest + slicing (device buffers, raw pointers) ----
    ORDER_REFERENCE, ORDER_FIRST_OCCURRENCE = 0, 1

    TIES_REFERENCE, TIES_SMALLER_PID = 0, 1

    def set_median_ties(self, mode):
        """What the composite entry points do at tied medians: "reference" (default) or "smaller_pid"."""
        code = {"reference": 0, "smaller_pid": 1, 0: 0, 1: 1}[mode]
        self._L.ecal_set_median_ties.argtypes = [ctypes.c_void_p, ctypes.c_int]
        self._check(self._L.ecal_set_median_ties(self._h, code))

    TAIL_AUTO, TAIL_TIERED, TAIL_LEAN = 0, 1, 2

    def set_tail_mode(self, mode):
        """How the stage calls schedule their later size tiers (ecal_set_tail_mode): "auto" (default: one general tail launch per
        stage while the stage's to-do lists were empty at its previous call), "tiered" (every tier, every call), "lean"."""
        code = {"auto": 0, "tiered": 1, "lean": 2, 0: 0, 1: 1, 2: 2}[mode]
        self._L.ecal_set_tail_mode.argtypes = [ctypes.c_void_p, ctypes.c_int]
        self._check(self._L.ecal_set_tail_mode(self._h, code))

    def get_tail_mode(self):
        """The mode in force as its number (TAIL_AUTO / TAIL_TIERED / TAIL_LEAN): set_tail_mode takes it back."""
        self._L.ecal_get_tail_mode.argtypes = [ctypes.c_void_p]
        self._L.ecal_get_tail_mode.restype = ctypes.c_int
        mode = self._L.ecal_get_tail_mode(self._h)
        if mode < 0:
            self._check(mode)
        return mode

    def set_profile_ranges(self, on=True):
        """roctx ranges around the stage entry points (ecal_set_profile_ranges) for `rocprofv3 --marker-trace --kernel-trace`."""
        self._L.ecal_set_profile_ranges.argtypes = [ctypes.c_void_p, ctypes.c_int]
        self._check(self._L.ecal_set_profile_ranges(self._h, 1 if on else 0))

    def set_point_order(self, order):
        """Element order of the pixel sets: "reference" (EventFrame.cpp:34-35 on libstdc++; default) or "first"."""
        code = {"reference": 0, "first": 1, 0: 0, 1: 1}[order]
        self._check(self._L.ecal_set_point_order(self._h, code))

    def point_order(self):
        return {0: "reference", 1: "first"}[self._L.ecal_get_point_order(self._h)]

    def window_bounds_dev(self, d_events, n_events, d_t0, d_t1, S, d_win_lo, d_win_hi, d_win_base, stream=0):
        self._check(self._L.ecal_window_bounds_dev(self._h, d_events, int(n_events), d_t0, d_t1, int(S), d_win_lo,
                                                   d_win_hi, d_win_base, stream))

    def sort_events_dev(self, d_events, n_events, d_sorted, stream=0):
        self._L.ecal_sort_events_dev.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
        self._L.ecal_sort_events_dev.restype = ctypes.c_int
        self._check(self._L.ecal_sort_events_dev(self._h, d_events, int(n_events), d_sorted, stream))

    def check_sorted_dev(self, d_events, n_events, d_flag, stream=0):
        self._check(self._L.ecal_check_sorted_dev(self._h, d_events, int(n_events), d_flag, stream))

    def slice_events_dev(self, d_events, n_events, d_win_lo, d_win_hi, d_win_base, S, max_win_events, cap_points,
                         d_xy, d_seg_off, d_seg_cnt, d_event_point, d_overflow, stream=0):
        self._check(self._L.ecal_slice_events_dev(self._h, d_events, int(n_events), d_win_lo, d_win_hi, d_win_base,
                                                  int(S), int(max_win_events), int(cap_points), d_xy, d_seg_off,
                                                  d_seg_cnt, d_event_point, d_overflow, stream))

    # ---- the three stages on packed points (ecal_packed_points) ----
    def slice_events_packed_dev(self, d_events, n_events, d_win_lo, d_win_hi, d_win_base, S, max_win_events, cap_points,
                                d_xy, d_seg_off, d_seg_cnt, d_event_point, d_overflow, pk, stream=0):
        L = self._L
        vp, u32 = ctypes.c_void_p, ctypes.c_uint32
        L.ecal_slice_events_packed_dev.argtypes = [vp, vp, ctypes.c_uint64, vp, vp, vp, u32, u32, u32, vp, vp, vp, vp, vp,
                                                   ctypes.POINTER(PackedPoints), vp]
        L.ecal_slice_events_packed_dev.restype = ctypes.c_int
        self._check(L.ecal_slice_events_packed_dev(self._h, d_events, int(n_events), d_win_lo, d_win_hi, d_win_base, int(S),
                                                   int(max_win_events), int(cap_points), d_xy, d_seg_off, d_seg_cnt, d_event_point,
                                                   d_overflow, ctypes.byref(pk) if pk is not None else None, stream))

    def dbscan_batch_packed_dev(self, d_xy, d_seg_off, d_seg_cnt, S, n_points, max_seg_points, eps, minpts, d_labels, d_n_clusters,
                                pk, stream=0):
        L = self._L
        vp, u32 = ctypes.c_void_p, ctypes.c_uint32
        L.ecal_dbscan_batch_packed_dev.argtypes = [vp, vp, vp, vp, u32, u32, u32, ctypes.c_double, u32, vp, vp,
                                                   ctypes.POINTER(PackedPoints), vp]
        L.ecal_dbscan_batch_packed_dev.restype = ctypes.c_int
        self._check(L.ecal_dbscan_batch_packed_dev(self._h, d_xy, d_seg_off, d_seg_cnt, int(S), int(n_points), int(max_seg_points),
                                                   float(eps), int(minpts), d_labels, d_n_clusters,
                                                   ctypes.byref(pk) if pk is not None else None, stream))

    def extract_batch_packed_dev(self, d_xy, d_seg_off, d_seg_cnt, d_labels, d_n_clusters, S, n_points, eps, cluster_min, need_clusters,
                                 radius_threshold, d_win_info, d_cand_pair, d_cand_xyr, d_kept_labels, d_rep, pk, stream=0,
                                 fit_circle=False, knn_num=3):
        """The extraction the context's median-ties setting asks for (exact by default), on packed points."""
        L = self._L
        vp, u32 = ctypes.c_void_p, ctypes.c_uint32
        L.ecal_extract_batch_packed_dev.argtypes = [vp, vp, vp, vp, vp, vp, u32, u32, ctypes.c_double, u32, u32, ctypes.c_double,
                                                    ctypes.c_int, u32, vp, vp, vp, vp, vp, ctypes.POINTER(PackedPoints), vp]
        L.ecal_extract_batch_packed_dev.restype = ctypes.c_int
        self._check(L.ecal_extract_batch_packed_dev(self._h, d_xy, d_seg_off, d_seg_cnt, d_labels, d_n_clusters, int(S), int(n_points),
                                                    float(eps), int(cluster_min), int(need_clusters), float(radius_threshold),
                                                    int(bool(fit_circle)), int(knn_num), d_win_info, d_cand_pair, d_cand_xyr,
                                                    d_kept_labels, d_rep, ctypes.byref(pk) if pk is not None else None, stream))

    def unpack_points_dev(self, pk, d_seg_off, d_seg_cnt, n_segments, d_xy, stream=0):
        L = self._L
        vp = ctypes.c_void_p
        L.ecal_unpack_points_dev.argtypes = [vp, ctypes.POINTER(PackedPoints), vp, vp, ctypes.c_uint32, vp, vp]
        L.ecal_unpack_points_dev.restype = ctypes.c_int
        self._check(L.ecal_unpack_points_dev(self._h, ctypes.byref(pk), d_seg_off, d_seg_cnt, int(n_segments), d_xy, stream))

    def get_median_ties(self):
        self._L.ecal_get_median_ties.argtypes = [ctypes.c_void_p]
        return int(self._L.ecal_get_median_ties(self._h))

    # ---- grid ordering ----
    def grid_order_dev(self, d_win_info, d_seg_off, d_cand_xyr, S, rows, cols, d_order, d_found, stream=0):
        self._check(self._L.ecal_grid_order_dev(self._h, d_win_info, d_seg_off, d_cand_xyr, int(S), int(rows), int(cols),
                                                d_order, d_found, stream))

    # ---- re-detection around predicted projections ----
    def rectify_batch_dev(self, d_xy, d_seg_off, d_seg_cnt, d_kept_labels, d_win_info, d_frame_window, d_pose, F,
                          d_landmarks, params, d_feat_xyr, d_feat_valid, d_frame_info, stream=0):
        self._check(self._L.ecal_rectify_batch_dev(self._h, d_xy, d_seg_off, d_seg_cnt, d_kept_labels, d_win_info,
                                                   d_frame_window, d_pose, int(F), d_landmarks, ctypes.byref(params),
                                                   d_feat_xyr, d_feat_valid, d_frame_info, stream))

    # ---- event -> residual association ----
    def associate_dev(self, d_events, n_events, d_kf_time, d_kf_circles, n_keyframes, n_circles, t_min, t_max,
                      max_dt, edge_tol, d_obs, d_time, d_lm_id, d_count, stream=0):
        self._check(self._L.ecal_associate_dev(self._h, d_events, int(n_events), d_kf_time, d_kf_circles,
                                               int(n_keyframes), int(n_circles), float(t_min), float(t_max),
                                               float(max_dt), float(edge_tol), d_obs, d_time, d_lm_id, d_count, stream))

    def associate_ranges_dev(self, d_events, n_events, d_kf_time, d_kf_circles, n_keyframes, n_circles, d_ranges, n_ranges,
                             max_dt, edge_tol, d_obs, d_time, d_lm_id, d_seg_id, d_count, stream=0):
        """ecal_associate_ranges_dev: all spline segments in one pass; the count stays on the device."""
        self._check(self._L.ecal_associate_ranges_dev(self._h, d_events, int(n_events), d_kf_time, d_kf_circles, int(n_keyframes),
                                                      int(n_circles), d_ranges, int(n_ranges), float(max_dt), float(edge_tol),
                                                      d_obs, d_time, d_lm_id, d_seg_id, d_count, stream))

    # ---- circle-candidate extraction ----
    def circle_radius_threshold(self, width, height, rows, cols, asymmetric, square_size, circle_radius):
        return self._L.ecal_circle_radius_threshold(float(width), float(height), int(rows), int(cols),
                                                    int(bool(asymmetric)), float(square_size), float(circle_radius))

    def extract_batch_dev(self, d_xy, d_seg_off, d_seg_cnt, d_labels, d_n_clusters, S, n_points, cluster_min,
                          need_clusters, radius_threshold, d_win_info, d_cand_pair, d_cand_xyr, d_kept_labels, d_rep,
                          stream=0, fit_circle=False, knn_num=3):
        self._check(self._L.ecal_extract_batch_dev(self._h, d_xy, d_seg_off, d_seg_cnt, d_labels, d_n_clusters,
                                                   int(S), int(n_points), int(cluster_min), int(need_clusters),
                                                   float(radius_threshold), int(bool(fit_circle)), int(knn_num),
                                                   d_win_info, d_cand_pair, d_cand_xyr, d_kept_labels, d_rep, stream))

    def extract_batch_ordered_dev(self, d_xy, d_seg_off, d_seg_cnt, d_labels, d_n_clusters, d_cluster_order, S, n_points, cluster_min,
                                  need_clusters, radius_threshold, d_win_info, d_cand_pair, d_cand_xyr, d_kept_labels, d_rep,
                                  stream=0, fit_circle=False, knn_num=3):
        """extract_batch_dev with the reference's own pick at tied medians (d_cluster_order from cluster_order_dev)."""
        L = self._L
        vp, u32 = ctypes.c_void_p, ctypes.c_uint32
        L.ecal_extract_batch_ordered_dev.argtypes = [vp, vp, vp, vp, vp, vp, vp, u32, u32, u32, u32, ctypes.c_double, ctypes.c_int, u32,
                                                     vp, vp, vp, vp, vp, vp]
        L.ecal_extract_batch_ordered_dev.restype = ctypes.c_int
        self._check(L.ecal_extract_batch_ordered_dev(self._h, d_xy, d_seg_off, d_seg_cnt, d_labels, d_n_clusters, d_cluster_order, int(S),
                                                     int(n_points), int(cluster_min), int(need_clusters), float(radius_threshold),
                                                     int(bool(fit_circle)), int(knn_num), d_win_info, d_cand_pair, d_cand_xyr,
                                                     d_kept_labels, d_rep, stream))

    def extract_batch_exact_dev(self, d_xy, d_seg_off, d_seg_cnt, d_labels, d_n_clusters, S, n_points, eps, cluster_min, need_clusters,
                                radius_threshold, d_win_info, d_cand_pair, d_cand_xyr, d_kept_labels, d_rep, stream=0, fit_circle=False,
                                knn_num=3):
        """The exact extraction in one call: plain pass + member order of the tied clusters + re-extraction of their windows."""
        L = self._L
        vp, u32 = ctypes.c_void_p, ctypes.c_uint32
        L.ecal_extract_batch_exact_dev.argtypes = [vp, vp, vp, vp, vp, vp, u32, u32, ctypes.c_double, u32, u32, ctypes.c_double,
                                                   ctypes.c_int, u32, vp, vp, vp, vp, vp, vp]
        L.ecal_extract_batch_exact_dev.restype = ctypes.c_int
        self._check(L.ecal_extract_batch_exact_dev(self._h, d_xy, d_seg_off, d_seg_cnt, d_labels, d_n_clusters, int(S), int(n_points),
                                                   float(eps), int(cluster_min), int(need_clusters), float(radius_threshold),
                                                   int(bool(fit_circle)), int(knn_num), d_win_info, d_cand_pair, d_cand_xyr,
                                                   d_kept_labels, d_rep, stream))

    def cluster_order_dev(self, d_xy, d_seg_off, d_seg_cnt, S, eps, d_labels, d_n_clusters, d_order, d_status, stream=0,
                          only_tied_medians=False):
        """Position of every core point inside the reference's Clusters[label] (expandCluster's pop order); -1 for noise.
        only_tied_medians: False / True, or 2 = only the clusters the caller marked with -3 in d_order (ecal.h)."""
        L = self._L
        vp = ctypes.c_void_p
        L.ecal_cluster_order_dev.argtypes = [vp, vp, vp, vp, ctypes.c_uint32, ctypes.c_double, vp, vp, vp, vp, ctypes.c_int, vp]
        L.ecal_cluster_order_dev.restype = ctypes.c_int
        self._check(L.ecal_cluster_order_dev(self._h, d_xy, d_seg_off, d_seg_cnt, int(S), float(eps), d_labels, d_n_clusters, d_order,
                                             d_status, int(only_tied_medians), stream))

    def cluster_order(self, xy, slice_off, eps, labels, n_clusters):
        """Host-buffer form of cluster_order_dev: returns (order [N] int32, status [S] uint32)."""
        L = self._L
        vp = ctypes.c_void_p
        xy = np.ascontiguousarray(xy, np.float64)
        slice_off = np.ascontiguousarray(slice_off, np.uint32)
        labels = np.ascontiguousarray(labels, np.int32)
        n_clusters = np.ascontiguousarray(n_clusters, np.uint32)
        S = len(slice_off) - 1
        order = np.full(max(len(labels), 1), -9, np.int32)
        status = np.full(max(S, 1), 9, np.uint32)
        L.ecal_cluster_order.argtypes = [vp, vp, vp, ctypes.c_uint32, ctypes.c_double, vp, vp, vp, vp]
        L.ecal_cluster_order.restype = ctypes.c_int
        self._check(L.ecal_cluster_order(self._h, _ptr(xy), _ptr(slice_off), S, float(eps), _ptr(labels), _ptr(n_clusters), _ptr(order),
                                         _ptr(status)))
        return order[:len(labels)], status[:S]

    def detect_fused_dev(self, d_events, n_events, d_win_lo, d_win_hi, d_win_base, S, max_win_events, max_seg_points, cap_points,
                         eps, minpts, cluster_min, need_clusters, radius_threshold, d_xy, d_seg_off, d_seg_cnt, d_event_point,
                         d_overflow, d_labels, d_n_clusters, d_win_info, d_cand_pair, d_cand_xyr, d_kept_labels, d_rep, stream=0,
                         fit_circle=False, knn_num=3):
        """slice + DBSCAN + extraction of every window in one call (one kernel in the shipped configuration)."""
        self._check(self._L.ecal_detect_fused_dev(self._h, d_events, int(n_events), d_win_lo, d_win_hi, d_win_base, int(S),
                                                  int(max_win_events), int(max_seg_points), int(cap_points), float(eps),
                                                  int(minpts), int(cluster_min), int(need_clusters), float(radius_threshold),
                                                  int(bool(fit_circle)), int(knn_num), d_xy, d_seg_off, d_seg_cnt, d_event_point,
                                                  d_overflow, d_labels, d_n_clusters, d_win_info, d_cand_pair, d_cand_xyr,
                                                  d_kept_labels, d_rep, stream))


# ---- continuous-time calibration solve ----
class _SplineProblem(ctypes.Structure):
    _fields_ = [("n_segments", ctypes.c_uint32), ("seg_cp_off", ctypes.c_void_p), ("knots", ctypes.c_void_p),
                ("n_res", ctypes.c_uint64), ("obs", ctypes.c_void_p), ("time", ctypes.c_void_p),
                ("lm_id", ctypes.c_void_p), ("seg_id", ctypes.c_void_p), ("n_landmarks", ctypes.c_uint32),
                ("landmarks", ctypes.c_void_p), ("circle_radius", ctypes.c_double), ("huber_a", ctypes.c_double),
                ("use_so3", ctypes.c_int), ("camera_model", ctypes.c_int)]
CAMERA_RADIAL, CAMERA_FISHEYE = 0, 1


ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)


class LmOptions(ctypes.Structure):
    _fields_ = [("max_num_iterations", ctypes.c_int), ("function_tolerance", ctypes.c_double),
                ("gradient_tolerance", ctypes.c_double), ("parameter_tolerance", ctypes.c_double),
                ("initial_trust_region_radius", ctypes.c_double), ("max_trust_region_radius", ctypes.c_double),
                ("min_relative_decrease", ctypes.c_double), ("min_lm_diagonal", ctypes.c_double),
                ("max_lm_diagonal", ctypes.c_double), ("jacobi_scaling", ctypes.c_int),
                ("allreduce", ALLREDUCE_FN), ("allreduce_user", ctypes.c_void_p),
                ("distributed", ctypes.c_int), ("rank", ctypes.c_int), ("world_size", ctypes.c_int)]


class LmSummary(ctypes.Structure):
    _fields_ = [("iterations", ctypes.c_int), ("successful_steps", ctypes.c_int), ("unsuccessful_steps", ctypes.c_int),
                ("jacobian_evaluations", ctypes.c_int), ("cost_evaluations", ctypes.c_int),
                ("termination", ctypes.c_int), ("initial_cost", ctypes.c_double), ("final_cost", ctypes.c_double),
                ("seconds", ctypes.c_double), ("seconds_evaluate", ctypes.c_double),
                ("seconds_linear_solve", ctypes.c_double)]


def _declare_solver(L):
    vp, i32, f64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
    L.ecal_solver_create.argtypes = [vp, ctypes.POINTER(_SplineProblem), ctypes.POINTER(vp)]
    L.ecal_solver_create.restype = i32
    L.ecal_solver_create_dev.argtypes = [vp, ctypes.POINTER(_SplineProblem), vp, vp, ctypes.POINTER(vp)]
    L.ecal_solver_create_dev.restype = i32
    L.ecal_solver_num_residuals.argtypes = [vp]
    L.ecal_solver_num_residuals.restype = ctypes.c_uint64
    L.ecal_solver_destroy.argtypes = [vp]
    L.ecal_solver_destroy.restype = None
    for f in (L.ecal_solver_param_size, L.ecal_solver_normal_size):
        f.argtypes = [vp]
        f.restype = ctypes.c_size_t
    L.ecal_solver_num_chunks.argtypes = [vp]
    L.ecal_solver_num_chunks.restype = ctypes.c_uint32
    L.ecal_solver_evaluate_dev.argtypes = [vp, vp, i32, vp, vp]
    L.ecal_solver_evaluate_dev.restype = i32
    L.ecal_solver_evaluate.argtypes = [vp, vp, i32, vp]
    L.ecal_solver_evaluate.restype = i32
    L.ecal_lm_default_options.argtypes = [ctypes.POINTER(LmOptions)]
    L.ecal_lm_default_options.restype = None
    L.ecal_solver_solve.argtypes = [vp, vp, ctypes.POINTER(LmOptions), ctypes.POINTER(LmSummary)]
    L.ecal_solver_solve.restype = i32
    L.ecal_inverse_radial_distortion.argtypes = [vp, vp]
    L.ecal_inverse_radial_distortion.restype = None


def time_shard_cuts(knots, n_cp, world_size):
    """ecal_solver_time_shard_cuts: the world_size - 1 cut times of a spline's time shards (LmOptions.distributed = 2)."""
    L = load_library()
    L.ecal_solver_time_shard_cuts.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p]
    L.ecal_solver_time_shard_cuts.restype = ctypes.c_int
    kn = np.ascontiguousarray(knots, np.float64)
    out = np.zeros(max(world_size - 1, 1))
    rc = L.ecal_solver_time_shard_cuts(kn.ctypes.data, int(n_cp), int(world_size), out.ctypes.data)
    if rc:
        raise EcalError(rc, "ecal_solver_time_shard_cuts")
    return out[:world_size - 1]


def pose_gates(Rsw, twb, time, pnp_ok, rect_ok, motion_time_step):
    """ecal_pose_gates: the sequential keyframe gates of EventCalibIni::cvCalibration (EventCalibIni.cpp:281-302) on per-frame
    results known for all frames.  Returns (accepted indices int64, discarded by checkPose, discarded by rectify)."""
    L = load_library()
    u8p, u32p, f64p = ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_double)
    L.ecal_pose_gates.argtypes = [ctypes.c_uint32, f64p, f64p, f64p, u8p, u8p, ctypes.c_double, u32p, u32p, u32p, u32p]
    L.ecal_pose_gates.restype = ctypes.c_int
    R = np.ascontiguousarray(Rsw, np.float64).reshape(-1, 9)
    K = R.shape[0]
    tw = np.ascontiguousarray(twb, np.float64).reshape(K, 3)
    tt = np.ascontiguousarray(time, np.float64).reshape(K)
    a = np.ascontiguousarray(pnp_ok, np.uint8).reshape(K)
    b = np.ascontiguousarray(rect_ok, np.uint8).reshape(K)
    acc = np.zeros(max(K, 1), np.uint32)
    n = (ctypes.c_uint32 * 3)()
    rc = L.ecal_pose_gates(K, R.ctypes.data_as(f64p), tw.ctypes.data_as(f64p), tt.ctypes.data_as(f64p), a.ctypes.data_as(u8p),
                           b.ctypes.data_as(u8p), float(motion_time_step), acc.ctypes.data_as(u32p),
                           ctypes.cast(ctypes.byref(n, 0), u32p), ctypes.cast(ctypes.byref(n, 4), u32p), ctypes.cast(ctypes.byref(n, 8), u32p))
    if rc:
        raise EcalError(rc, "ecal_pose_gates: invalid arguments")
    return acc[:n[0]].astype(np.int64), int(n[1]), int(n[2])


def inverse_radial_distortion(k4):
    L = load_library()
    _declare_solver(L)
    k = np.ascontiguousarray(k4, np.float64)
    b = np.zeros(5)
    L.ecal_inverse_radial_distortion(_ptr(k), _ptr(b))
    return b


class Solver:
    """ecal_solver: residual records + spline layout resident on the GPU.

    problem: dict with seg_cp_off [G+1] u32, knots f64, obs [M,2], time [M], lm_id [M] u32,
    seg_id [M] u32 or None, landmarks [L,3], circle_radius, huber_a, use_so3 / fisheye (optional, default False).
    Parameter vector layout: [intr 9 | q n_cp x 4 (xyzw) | t n_cp x 3]."""

    def __init__(self, ctx: Context, problem, device_arrays=None, stream=0):
        """device_arrays = (d_obs, d_time, d_lm_id, d_seg_id or None, capacity, d_count or None): raw DEVICE pointers of the
        residual arrays (ecal_associate_ranges_dev's outputs) — ecal_solver_create_dev builds the problem in place; the
        dict then carries only seg_cp_off, knots, landmarks and the scalars."""
        self.ctx = ctx
        L = ctx._L
        _declare_solver(L)
        keep = {}

        def arr(name, dt):
            a = np.ascontiguousarray(problem[name], dtype=dt)
            keep[name] = a
            return ctypes.c_void_p(a.ctypes.data)

        P = _SplineProblem()
        P.seg_cp_off = arr("seg_cp_off", np.uint32)
        P.n_segments = keep["seg_cp_off"].shape[0] - 1
        P.knots = arr("knots", np.float64)
        if device_arrays is None:
            P.obs = arr("obs", np.float64)
            P.time = arr("time", np.float64)
            P.lm_id = arr("lm_id", np.uint32)
            P.n_res = keep["time"].shape[0]
            P.seg_id = arr("seg_id", np.uint32) if problem.get("seg_id") is not None else None
        else:
            d_obs, d_time, d_lm, d_seg, cap, d_count = device_arrays
            P.obs, P.time, P.lm_id, P.seg_id, P.n_res = d_obs, d_time, d_lm, d_seg, int(cap)
        P.landmarks = arr("landmarks", np.float64)
        P.n_landmarks = keep["landmarks"].reshape(-1, 3).shape[0]
        P.circle_radius = float(problem["circle_radius"])
        P.huber_a = float(problem["huber_a"])
        P.use_so3 = int(bool(problem.get("use_so3", False)))
        P.camera_model = CAMERA_FISHEYE if problem.get("fisheye", False) else CAMERA_RADIAL
        h = ctypes.c_void_p()
        if device_arrays is None:
            ctx._check(L.ecal_solver_create(ctx._h, ctypes.byref(P), ctypes.byref(h)))
        else:
            ctx._check(L.ecal_solver_create_dev(ctx._h, ctypes.byref(P), device_arrays[5], stream, ctypes.byref(h)))
        self._h = h
        self.n_params = int(L.ecal_solver_param_size(h))
        self.n_normal = int(L.ecal_solver_normal_size(h))
        self.n_cp = (self.n_params - 9) // 7
        self.n_chunks = int(L.ecal_solver_num_chunks(h))
        self.n_res = int(L.ecal_solver_num_residuals(h))

    def close(self):
        if getattr(self, "_h", None):
            self.ctx._L.ecal_solver_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def evaluate(self, params, with_jacobian=True):
        p = np.ascontiguousarray(params, np.float64)
        assert p.shape[0] == self.n_params
        acc = np.zeros(self.n_normal if with_jacobian else 1)
        self.ctx._check(self.ctx._L.ecal_solver_evaluate(self._h, _ptr(p), int(with_jacobian), _ptr(acc)))
        return acc

    def residuals(self, params, with_jacobian=True):
        """ecal_residuals: (r [n_res], J [n_res, 33] or None, cp0 [n_res]) — the per-residual CostFunction::Evaluate seam."""
        p = np.ascontiguousarray(params, np.float64)
        assert p.shape[0] == self.n_params
        L = self.ctx._L
        L.ecal_residuals.argtypes = [ctypes.c_void_p] * 5
        L.ecal_residuals.restype = ctypes.c_int
        n = int(self.n_res)
        r = np.zeros(max(n, 1))
        J = np.zeros((max(n, 1), 33)) if with_jacobian else None
        c = np.zeros(max(n, 1), np.uint32)
        self.ctx._check(L.ecal_residuals(self._h, _ptr(p), _ptr(r), _ptr(J) if with_jacobian else None, _ptr(c)))
        return r[:n], (J[:n] if with_jacobian else None), c[:n]

    def evaluate_dev(self, d_params, with_jacobian, d_accum, stream=0):
        self.ctx._check(self.ctx._L.ecal_solver_evaluate_dev(self._h, d_params, int(with_jacobian), d_accum, stream))

    def default_options(self):
        o = LmOptions()
        self.ctx._L.ecal_lm_default_options(ctypes.byref(o))
        return o

    def solve(self, params, options=None):
        p = np.array(params, dtype=np.float64, copy=True)
        s = LmSummary()
        o = options if options is not None else self.default_options()
        self.ctx._check(self.ctx._L.ecal_solver_solve(self._h, _ptr(p), ctypes.byref(o), ctypes.byref(s)))
        return p, s


def unpack_normal(acc, n_cp):
    """Normal-equation buffer -> (cost, g [9+6 n_cp], H dense symmetric) in the order [intr | cp0 (rot3, trans3) | ...]."""
    n = 9 + 6 * n_cp
    g = np.zeros(n)
    H = np.zeros((n, n))
    g[:9] = acc[1:10]
    Hi = acc[10:91].reshape(9, 9)
    H[:9, :9] = np.triu(Hi) + np.triu(Hi, 1).T
    for c in range(n_cp):
        b = acc[91 + 204 * c: 91 + 204 * (c + 1)]
        r0 = 9 + 6 * c
        g[r0:r0 + 6] = b[:6]
        H[r0:r0 + 6, :9] = b[6:60].reshape(6, 9)
        H[:9, r0:r0 + 6] = H[r0:r0 + 6, :9].T
        for d in range(4):
            if c + d >= n_cp:
                break
            blk = b[60 + 36 * d: 96 + 36 * d].reshape(6, 6)
            c1 = 9 + 6 * (c + d)
            if d == 0:
                H[r0:r0 + 6, r0:r0 + 6] = np.triu(blk) + np.triu(blk, 1).T
            else:
                H[r0:r0 + 6, c1:c1 + 6] = blk
                H[c1:c1 + 6, r0:r0 + 6] = blk.T
    return acc[0], g, H


def make_allreduce_hook(ctx: Context, world_size):
    """ALLREDUCE_FN for LmOptions.allreduce: sums the solver's device buffer over ranks with
    torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" works for CPU-side tests of the
    plumbing).  The buffer is staged through a torch tensor; full synchronisation on both sides keeps the
    solver's private stream and torch's collective stream ordered."""
    import torch
    import torch.distributed as dist
    state = {}

    def hook(user, d_buf, n, stream):
        try:
            if world_size <= 1:
                return 0
            t = state.get(n)
            if t is None:
                t = state[n] = torch.empty(n, dtype=torch.float64, device=torch.device("cuda", ctx.device))
            ctx._check(ctx._L.ecal_copy_dev(ctx._h, t.data_ptr(), d_buf, n * 8, stream, 1))
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            torch.cuda.synchronize(t.device)
            ctx._check(ctx._L.ecal_copy_dev(ctx._h, d_buf, t.data_ptr(), n * 8, stream, 1))
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            print("allreduce hook failed:", e)
            return 1

    return ALLREDUCE_FN(hook)


# ---- init calibration on calibration views ----
CALIB_FIX_ASPECT_RATIO, CALIB_FIX_PRINCIPAL_POINT, CALIB_ZERO_TANGENT_DIST = 1 << 0, 1 << 1, 1 << 2
CALIB_FIX_K1, CALIB_FIX_K2, CALIB_FIX_K3, CALIB_FIX_K4, CALIB_FIX_K5, CALIB_FIX_K6 = (1 << 3, 1 << 4, 1 << 5, 1 << 6,
                                                                                      1 << 7, 1 << 8)
CALIB_FIX_SKEW, CALIB_RECOMPUTE_EXTRINSIC = 1 << 9, 1 << 10
CALIB_USE_INTRINSIC_GUESS = 1 << 11
CALIB_BLOCK_DOUBLES = 272


class CalibOptions(ctypes.Structure):
    """ecal_calib_options (include/ecal.h)."""
    _fields_ = [("model", ctypes.c_int), ("flags", ctypes.c_uint32), ("aspect_ratio", ctypes.c_double),
                ("max_iter", ctypes.c_int), ("eps", ctypes.c_double), ("allreduce", ALLREDUCE_FN),
                ("allreduce_user", ctypes.c_void_p)]


class CalibResult(ctypes.Structure):
    _fields_ = [("intr", ctypes.c_double * 12), ("rms", ctypes.c_double), ("iterations", ctypes.c_int),
                ("jacobian_evaluations", ctypes.c_int), ("error_evaluations", ctypes.c_int), ("seconds", ctypes.c_double)]


def _declare_calib(L):
    vp, i32, u32, f64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_double
    L.ecal_calib_default_options.argtypes = [ctypes.POINTER(CalibOptions)]
    L.ecal_calib_default_options.restype = None
    L.ecal_calib_view_blocks_dev.argtypes = [vp, vp, u32, vp, u32, i32, u32, f64, vp, vp, i32, vp, vp]
    L.ecal_calib_view_blocks_dev.restype = i32
    L.ecal_pnp_batch_dev.argtypes = [vp, vp, u32, vp, vp, u32, i32, vp, f64, i32, i32, vp, vp, vp, vp, vp]
    L.ecal_pnp_batch_dev.restype = i32
    L.ecal_calibrate_views.argtypes = [vp, vp, u32, vp, u32, f64, f64, ctypes.POINTER(CalibOptions),
                                       ctypes.POINTER(CalibResult), vp, vp, vp]
    L.ecal_calibrate_views.restype = i32


def calibrate_views(ctx: Context, obj, img, width, height, model=0, flags=0, aspect_ratio=1.0, max_iter=0, eps=0.0,
                    allreduce=None, intr_guess=None):
    """ecal_calibrate_views: obj [n][3], img [V][n][2] (this rank's views).  Returns a dict.  intr_guess [12] goes with
    CALIB_USE_INTRINSIC_GUESS in flags."""
    L = ctx._L
    _declare_calib(L)
    obj = np.ascontiguousarray(obj, np.float64)
    img = np.ascontiguousarray(img, np.float64).reshape(-1, obj.shape[0], 2)
    V = img.shape[0]
    opt = CalibOptions()
    L.ecal_calib_default_options(ctypes.byref(opt))
    opt.model, opt.flags, opt.aspect_ratio, opt.max_iter, opt.eps = int(model), int(flags), float(aspect_ratio), int(max_iter), float(eps)
    if allreduce is not None:
        if isinstance(allreduce, tuple):   # Context.comm_allreduce_fn(): (function, user pointer)
            opt.allreduce, opt.allreduce_user = allreduce
        else:
            opt.allreduce = allreduce
    res = CalibResult()
    if intr_guess is not None:
        for j, v in enumerate(np.asarray(intr_guess, np.float64)[:12]):
            res.intr[j] = float(v)
    rv, tv, pe = np.zeros((V, 3)), np.zeros((V, 3)), np.zeros(V)
    ctx._check(L.ecal_calibrate_views(ctx._h, _ptr(obj), obj.shape[0], _ptr(img) if V else None, V, float(width), float(height),
                                      ctypes.byref(opt), ctypes.byref(res), _ptr(rv) if V else None, _ptr(tv) if V else None,
                                      _ptr(pe) if V else None))
    return {"intr": np.array(res.intr[:]), "rms": res.rms, "iterations": res.iterations, "rvecs": rv, "tvecs": tv,
            "per_view_err": pe, "jacobian_evaluations": res.jacobian_evaluations,
            "error_evaluations": res.error_evaluations, "seconds": res.seconds}


def calibrate_fisheye_views(ctx: Context, obj, img, width, height, flags=0, aspect_ratio=1.0, allreduce=None):
    """ecal_calibrate_fisheye_views: the fisheye init calibration with the library's start procedure (the reference's own start
    first, the radial model's focal lengths as a guess when that fails).  Returns calibrate_views' dict + "start_used"."""
    L = ctx._L
    _declare_calib(L)
    obj = np.ascontiguousarray(obj, np.float64)
    img = np.ascontiguousarray(img, np.float64).reshape(-1, obj.shape[0], 2)
    V = img.shape[0]
    opt = CalibOptions()
    L.ecal_calib_default_options(ctypes.byref(opt))
    opt.model, opt.flags, opt.aspect_ratio = 1, int(flags), float(aspect_ratio)
    if allreduce is not None:
        if isinstance(allreduce, tuple):
            opt.allreduce, opt.allreduce_user = allreduce
        else:
            opt.allreduce = allreduce
    res = CalibResult()
    rv, tv, pe = np.zeros((V, 3)), np.zeros((V, 3)), np.zeros(V)
    used = ctypes.c_int(-1)
    L.ecal_calibrate_fisheye_views.restype = ctypes.c_int
    L.ecal_calibrate_fisheye_views.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32,
                                               ctypes.c_double, ctypes.c_double, ctypes.POINTER(CalibOptions), ctypes.POINTER(CalibResult),
                                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    ctx._check(L.ecal_calibrate_fisheye_views(ctx._h, _ptr(obj), obj.shape[0], _ptr(img) if V else None, V, float(width), float(height),
                                              ctypes.byref(opt), ctypes.byref(res), _ptr(rv) if V else None, _ptr(tv) if V else None,
                                              _ptr(pe) if V else None, ctypes.byref(used)))
    return {"intr": np.array(res.intr[:]), "rms": res.rms, "iterations": res.iterations, "rvecs": rv, "tvecs": tv,
            "per_view_err": pe, "jacobian_evaluations": res.jacobian_evaluations, "error_evaluations": res.error_evaluations,
            "seconds": res.seconds, "start_used": int(used.value)}


def calib_view_blocks_dev(ctx: Context, d_obj, n_pts, d_img, n_views, model, flags, aspect_ratio, d_intr, d_view_params,
                          with_jacobian, d_blocks, stream=0):
    _declare_calib(ctx._L)
    ctx._check(ctx._L.ecal_calib_view_blocks_dev(ctx._h, d_obj, n_pts, d_img, n_views, model, flags, aspect_ratio, d_intr,
                                                 d_view_params, int(with_jacobian), d_blocks, stream))


def pnp_batch_dev(ctx: Context, d_obj, n_pts, d_img, d_valid, n_frames, model, d_intr, reproj_thresh, rounds, refine_iters,
                  d_pose, d_inlier=None, d_err=None, d_ok=None, stream=0):
    _declare_calib(ctx._L)
    ctx._check(ctx._L.ecal_pnp_batch_dev(ctx._h, d_obj, n_pts, d_img, d_valid, n_frames, model, d_intr, float(reproj_thresh),
                                         int(rounds), int(refine_iters), d_pose, d_inlier, d_err, d_ok, stream))


def spline_fit(u, data, n_cp):
    """ecal_spline_fit: returns (knots [n_cp + 4], control points [n_cp][dim])."""
    L = load_library()
    u = np.ascontiguousarray(u, np.float64)
    data = np.ascontiguousarray(data, np.float64).reshape(len(u), -1)
    knots, cp = np.zeros(n_cp + 4), np.zeros((n_cp, data.shape[1]))
    L.ecal_spline_fit.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                  ctypes.c_void_p, ctypes.c_void_p]
    st = L.ecal_spline_fit(_ptr(u), _ptr(data), len(u), data.shape[1], n_cp, _ptr(knots), _ptr(cp))
    if st != 0:
        raise EcalError(st, L.ecal_strerror(st).decode())
    return knots, cp


def spline_eval(knots, cp, u):
    L = load_library()
    knots, cp, u = (np.ascontiguousarray(a, np.float64) for a in (knots, cp, u))
    out = np.zeros((len(u), cp.shape[1]))
    L.ecal_spline_eval.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p,
                                   ctypes.c_uint32, ctypes.c_void_p]
    st = L.ecal_spline_eval(_ptr(knots), _ptr(cp), cp.shape[0], cp.shape[1], _ptr(u), len(u), _ptr(out))
    if st != 0:
        raise EcalError(st, L.ecal_strerror(st).decode())
    return out


def spline_so3_refine(knots, cp_quat, sample_quat, u, max_iterations=0):
    """ecal_spline_so3_refine (BsplineSO3::optimizeCP): returns (refined control points [n_cp][4], summary dict)."""
    L = load_library()
    knots, u = np.ascontiguousarray(knots, np.float64), np.ascontiguousarray(u, np.float64)
    cp = np.ascontiguousarray(cp_quat, np.float64).copy()
    sq = np.ascontiguousarray(sample_quat, np.float64)
    c0, c1, it = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_int(0)
    L.ecal_spline_so3_refine.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(ctypes.c_double),
                                         ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]
    st = L.ecal_spline_so3_refine(_ptr(knots), cp.shape[0], _ptr(cp), _ptr(sq), _ptr(u), len(u), int(max_iterations),
                                  ctypes.byref(c0), ctypes.byref(c1), ctypes.byref(it))
    if st != 0:
        raise EcalError(st, L.ecal_strerror(st).decode())
    return cp, {"initial_cost": c0.value, "final_cost": c1.value, "iterations": it.value}


# ---- double-buffered ingest ----
class DetectParams(ctypes.Structure):
    """ecal_detect_params (include/ecal.h)."""
    _fields_ = [("dbscan_eps", ctypes.c_double), ("dbscan_min_samples", ctypes.c_uint32), ("cluster_min_sample", ctypes.c_uint32),
                ("need_clusters", ctypes.c_uint32), ("circle_radius_threshold", ctypes.c_double), ("fit_circle", ctypes.c_int),
                ("knn_num", ctypes.c_uint32), ("rows", ctypes.c_uint32), ("cols", ctypes.c_uint32)]


class IngestStats(ctypes.Structure):
    _fields_ = [("chunks", ctypes.c_uint32), ("max_chunk_events", ctypes.c_uint64), ("bytes_uploaded", ctypes.c_uint64),
                ("seconds", ctypes.c_double)]


def detect_stream_tiled(ctx: Context, host_ptr, n_events, t_start, window_len, windows_per_chunk, max_windows, eps=4.0, minpts=2,
                        cluster_min=5, rows=9, cols=4, radius_threshold=15.511363636363637, want_features=True):
    """ecal_detect_stream_tiled over packed records at host address `host_ptr` (pinned for overlap).  Returns
    (win_info [S,4], grid_found [S], features [S, rows*cols, 3] or None, stats dict)."""
    L = ctx._L
    vp, u32, f64 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_double
    L.ecal_detect_stream_tiled.argtypes = [vp, vp, ctypes.c_uint64, f64, f64, u32, ctypes.POINTER(DetectParams), u32, vp, vp, vp,
                                           ctypes.POINTER(u32), ctypes.POINTER(IngestStats)]
    L.ecal_detect_stream_tiled.restype = ctypes.c_int
    prm = DetectParams(float(eps), int(minpts), int(cluster_min), int(rows * cols), float(radius_threshold), 0, 3, int(rows), int(cols))
    info = np.zeros((max_windows, 4), np.uint32)
    found = np.zeros(max_windows, np.uint32)
    feat = np.zeros((max_windows, rows * cols, 3)) if want_features else None
    nw = u32(0)
    st = IngestStats()
    ctx._check(L.ecal_detect_stream_tiled(ctx._h, host_ptr, int(n_events), float(t_start), float(window_len), int(windows_per_chunk),
                                          ctypes.byref(prm), int(max_windows), _ptr(info), _ptr(found),
                                          _ptr(feat) if want_features else None, ctypes.byref(nw), ctypes.byref(st)))
    S = nw.value
    return info[:S], found[:S], (feat[:S] if want_features else None), {"chunks": st.chunks, "max_chunk_events": st.max_chunk_events,
                                                                      "bytes_uploaded": st.bytes_uploaded, "seconds": st.seconds}


class AdaptiveParams(ctypes.Structure):
    _fields_ = [("motion_time_step", ctypes.c_double), ("frame_event_num_threshold", ctypes.c_uint32), ("piece_num", ctypes.c_uint32),
                ("start_time", ctypes.c_double), ("end_time", ctypes.c_double), ("max_passes", ctypes.c_uint32),
                ("check_every", ctypes.c_uint32), ("gate_mode", ctypes.c_int), ("piece_first", ctypes.c_uint32),
                ("piece_count", ctypes.c_uint32)]
GATE_OWN_PIECE, GATE_SHARED_MAP = 0, 1


def detect_keyframes_cap_hint(ctx: Context, n_events, motion_time_step, frame_event_num_threshold, piece_num, start_time, end_time,
                              piece_first=0, piece_count=0):
    """ecal_detect_keyframes_cap_hint: a cap_points that detect_keyframes_dev will usually find sufficient."""
    L = ctx._L
    L.ecal_detect_keyframes_cap_hint.argtypes = [ctypes.POINTER(AdaptiveParams), ctypes.c_uint64]
    L.ecal_detect_keyframes_cap_hint.restype = ctypes.c_uint64
    ap = AdaptiveParams(float(motion_time_step), int(frame_event_num_threshold), int(piece_num), float(start_time), float(end_time), 0, 0, 0,
                        int(piece_first), int(piece_count))
    return int(L.ecal_detect_keyframes_cap_hint(ctypes.byref(ap), int(n_events)))


def detect_keyframes_cap_hint_dev(ctx: Context, d_events, n_events, motion_time_step, frame_event_num_threshold, piece_num, start_time,
                                  end_time, piece_first=0, piece_count=0):
    """ecal_detect_keyframes_cap_hint_dev: the hint from the events of the resident stream inside [start_time, end_time]."""
    L = ctx._L
    L.ecal_detect_keyframes_cap_hint_dev.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(AdaptiveParams)]
    L.ecal_detect_keyframes_cap_hint_dev.restype = ctypes.c_uint64
    ap = AdaptiveParams(float(motion_time_step), int(frame_event_num_threshold), int(piece_num), float(start_time), float(end_time), 0, 0, 0,
                        int(piece_first), int(piece_count))
    return int(L.ecal_detect_keyframes_cap_hint_dev(ctx._h, d_events, int(n_events), ctypes.byref(ap)))


class KeyframeFrame(ctypes.Structure):   # ecal_keyframe_frame
    _fields_ = [("has", ctypes.c_int), ("time", ctypes.c_double), ("dir", ctypes.c_double * 64)]


_FRAME_RECV = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(KeyframeFrame), ctypes.c_int)
_FRAME_SEND = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(KeyframeFrame))


class AdaptiveHandover(ctypes.Structure):   # ecal_adaptive_handover
    _fields_ = [("recv", _FRAME_RECV), ("send", _FRAME_SEND), ("user", ctypes.c_void_p)]


def detect_keyframes_dev(ctx: Context, d_events, n_events, motion_time_step, frame_event_num_threshold, piece_num, start_time,
                         end_time, cap_points, max_keyframes, eps=4.0, minpts=2, cluster_min=5, rows=9, cols=4,
                         radius_threshold=15.511363636363637, max_passes=0, check_every=0, gate_mode=0, piece_first=0, piece_count=0,
                         handover=None):
    """ecal_detect_keyframes (policy on the device; gate_mode: GATE_OWN_PIECE / GATE_SHARED_MAP; piece_count != 0: only the pieces
    piece_first .. piece_first + piece_count - 1 of the piece_num).  Returns (time [K], duration [K,2], events_num [K], features [K, rows cols, 3],
    passes, windows); raises EcalError(-6) when cap_points or max_keyframes is too small.
    handover (ecal_detect_keyframes_sharded: a subset of the pieces under GATE_SHARED_MAP): an object with recv(wait) -> None (not there
    yet; only when wait is false) or (has, time, dir [2 rows]) = the frame behind the pieces before this subset, and send(has, time,
    dir) for the frame behind this subset's pieces."""
    L = ctx._L
    vp, u32 = ctypes.c_void_p, ctypes.c_uint32
    L.ecal_detect_keyframes.argtypes = [vp, vp, ctypes.c_uint64, ctypes.POINTER(AdaptiveParams), ctypes.POINTER(DetectParams), u32, u32,
                                        vp, vp, vp, vp, ctypes.POINTER(u32), ctypes.POINTER(u32), ctypes.POINTER(ctypes.c_uint64)]
    L.ecal_detect_keyframes.restype = ctypes.c_int
    ap = AdaptiveParams(float(motion_time_step), int(frame_event_num_threshold), int(piece_num), float(start_time), float(end_time),
                        int(max_passes), int(check_every), int(gate_mode), int(piece_first), int(piece_count))
    prm = DetectParams(float(eps), int(minpts), int(cluster_min), int(rows * cols), float(radius_threshold), 0, 3, int(rows), int(cols))
    M = rows * cols
    t = np.empty(max_keyframes)
    d = np.empty((max_keyframes, 2))
    e = np.empty(max_keyframes, np.int32)
    f = np.empty((max_keyframes, M, 3))
    nk, ps, nw = u32(0), u32(0), ctypes.c_uint64(0)
    try:
        if handover is None:
            ctx._check(L.ecal_detect_keyframes(ctx._h, d_events, int(n_events), ctypes.byref(ap), ctypes.byref(prm), int(cap_points),
                                               int(max_keyframes), _ptr(t), _ptr(d), _ptr(e), _ptr(f), ctypes.byref(nk), ctypes.byref(ps),
                                               ctypes.byref(nw)))
        else:
            errs = []

            def recv(user, frame, wait):
                try:
                    got = handover.recv(bool(wait))
                    if got is None:
                        return 0
                    frame[0].has, frame[0].time = int(bool(got[0])), float(got[1])
                    for i, v in enumerate(got[2][: 2 * rows]):
                        frame[0].dir[i] = float(v)
                    return 1
                except BaseException as ex:   # noqa: BLE001 — never let an exception cross the C frame
                    errs.append(ex)
                    return -1

            def send(user, frame):
                try:
                    handover.send(int(frame[0].has), float(frame[0].time), [float(frame[0].dir[i]) for i in range(2 * rows)])
                    return 0
                except BaseException as ex:   # noqa: BLE001
                    errs.append(ex)
                    return -1
            ho = AdaptiveHandover(_FRAME_RECV(recv), _FRAME_SEND(send), None)
            L.ecal_detect_keyframes_sharded.argtypes = L.ecal_detect_keyframes.argtypes + [ctypes.POINTER(AdaptiveHandover)]
            L.ecal_detect_keyframes_sharded.restype = ctypes.c_int
            st = L.ecal_detect_keyframes_sharded(ctx._h, d_events, int(n_events), ctypes.byref(ap), ctypes.byref(prm), int(cap_points),
                                                 int(max_keyframes), _ptr(t), _ptr(d), _ptr(e), _ptr(f), ctypes.byref(nk), ctypes.byref(ps),
                                                 ctypes.byref(nw), ctypes.byref(ho))
            if errs:
                raise errs[0]
            ctx._check(st)
    except EcalError as err:
        err.n_keyframes = int(nk.value)     # ECAL_ERR_RANGE with a count beyond max_keyframes: the keyframe capacity was short
        raise
    K = nk.value
    return t[:K].copy(), d[:K].copy(), e[:K].astype(np.int64), f[:K].copy(), ps.value, nw.value


def detect_pass(ctx: Context, d_events, n_events, t0, t1, cap_points, eps=4.0, minpts=2, cluster_min=5, rows=9, cols=4,
                radius_threshold=15.511363636363637):
    """ecal_detect_pass: packed [S, 3 + 3 rows cols] = status, ok, unique pixels, ordered circles of every window."""
    L = ctx._L
    vp, u32 = ctypes.c_void_p, ctypes.c_uint32
    L.ecal_detect_pass.argtypes = [vp, vp, ctypes.c_uint64, vp, vp, u32, ctypes.POINTER(DetectParams), u32, vp]
    L.ecal_detect_pass.restype = ctypes.c_int
    t0 = np.ascontiguousarray(t0, np.float64)
    t1 = np.ascontiguousarray(t1, np.float64)
    S = t0.shape[0]
    prm = DetectParams(float(eps), int(minpts), int(cluster_min), int(rows * cols), float(radius_threshold), 0, 3, int(rows), int(cols))
    out = np.empty((S, 3 + 3 * rows * cols))
    ctx._check(L.ecal_detect_pass(ctx._h, d_events, int(n_events), _ptr(t0), _ptr(t1), S, ctypes.byref(prm), int(cap_points), _ptr(out)))
    return out
