"""ctypes binding of libecal.so (include/ecal.h).  No torch import here: device pointers are
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
    "ecal_window_bounds_dev", "ecal_check_sorted_dev", "ecal_slice_events_dev",
    "ecal_circle_radius_threshold", "ecal_extract_batch_dev",
    "ecal_stream_create", "ecal_stream_destroy", "ecal_stream_size", "ecal_detect_batch",
]


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
    L.ecal_window_bounds_dev.argtypes = [vp, vp, u64, vp, vp, u32, vp, vp, vp, vp]
    L.ecal_window_bounds_dev.restype = i32
    L.ecal_check_sorted_dev.argtypes = [vp, vp, u64, vp, vp]
    L.ecal_check_sorted_dev.restype = i32
    L.ecal_slice_events_dev.argtypes = [vp, vp, u64, vp, vp, vp, u32, u32, u32, vp, vp, vp, vp, vp, vp]
    L.ecal_slice_events_dev.restype = i32
    L.ecal_circle_radius_threshold.argtypes = [f64, f64, i32, i32, i32, f64, f64]
    L.ecal_circle_radius_threshold.restype = f64
    L.ecal_extract_batch_dev.argtypes = [vp, vp, vp, vp, vp, vp, u32, u32, u32, u32, f64, vp, vp, vp, vp, vp, vp]
    L.ecal_extract_batch_dev.restype = i32
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

    def close(self):
        if getattr(self, "_h", None):
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

    # ---- ingest + slicing (device buffers, raw pointers) ----
    def window_bounds_dev(self, d_events, n_events, d_t0, d_t1, S, d_win_lo, d_win_hi, d_win_base, stream=0):
        self._check(self._L.ecal_window_bounds_dev(self._h, d_events, int(n_events), d_t0, d_t1, int(S), d_win_lo,
                                                   d_win_hi, d_win_base, stream))

    def check_sorted_dev(self, d_events, n_events, d_flag, stream=0):
        self._check(self._L.ecal_check_sorted_dev(self._h, d_events, int(n_events), d_flag, stream))

    def slice_events_dev(self, d_events, n_events, d_win_lo, d_win_hi, d_win_base, S, max_win_events, cap_points,
                         d_xy, d_seg_off, d_seg_cnt, d_event_point, d_overflow, stream=0):
        self._check(self._L.ecal_slice_events_dev(self._h, d_events, int(n_events), d_win_lo, d_win_hi, d_win_base,
                                                  int(S), int(max_win_events), int(cap_points), d_xy, d_seg_off,
                                                  d_seg_cnt, d_event_point, d_overflow, stream))

    # ---- circle-candidate extraction ----
    def circle_radius_threshold(self, width, height, rows, cols, asymmetric, square_size, circle_radius):
        return self._L.ecal_circle_radius_threshold(float(width), float(height), int(rows), int(cols),
                                                    int(bool(asymmetric)), float(square_size), float(circle_radius))

    def extract_batch_dev(self, d_xy, d_seg_off, d_seg_cnt, d_labels, d_n_clusters, S, n_points, cluster_min,
                          need_clusters, radius_threshold, d_win_info, d_cand_pair, d_cand_xyr, d_kept_labels, d_rep,
                          stream=0):
        self._check(self._L.ecal_extract_batch_dev(self._h, d_xy, d_seg_off, d_seg_cnt, d_labels, d_n_clusters,
                                                   int(S), int(n_points), int(cluster_min), int(need_clusters),
                                                   float(radius_threshold), d_win_info, d_cand_pair, d_cand_xyr,
                                                   d_kept_labels, d_rep, stream))
