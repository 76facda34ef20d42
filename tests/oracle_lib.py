"""ctypes access to the CPU oracle (oracle/liboracle.so) — test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_KDTREE_SO = os.path.join(ORACLE_DIR, "_ref", "libkdtree_ref.so")

_lib = None


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        _lib = ctypes.CDLL(ORACLE_SO)
        _declare(_lib)
    return _lib


def have_ref_kdtree():
    return os.path.exists(REF_KDTREE_SO)


_dp = ctypes.POINTER(ctypes.c_double)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_i32p = ctypes.POINTER(ctypes.c_int32)
_u64p = ctypes.POINTER(ctypes.c_uint64)


def _declare(L):
    L.oracle_dbscan.argtypes = [_dp, ctypes.c_uint32, ctypes.c_double, ctypes.c_uint32, _i32p, _u32p, _u32p, _u32p]
    L.oracle_dbscan.restype = ctypes.c_int
    L.oracle_dbscan_kdapi.argtypes = [ctypes.c_char_p] + L.oracle_dbscan.argtypes
    L.oracle_dbscan_kdapi.restype = ctypes.c_int
    L.oracle_range_query_all.argtypes = [_dp, ctypes.c_uint32, ctypes.c_double, _u64p, _u32p, ctypes.c_uint64]
    L.oracle_range_query_all.restype = ctypes.c_long
    L.oracle_range_query_all_kdapi.argtypes = [ctypes.c_char_p] + L.oracle_range_query_all.argtypes
    L.oracle_range_query_all_kdapi.restype = ctypes.c_long
    L.oracle_dbscan_batch.argtypes = [_dp, _u32p, _u32p, ctypes.c_uint32, ctypes.c_double, ctypes.c_uint32, _i32p, _u32p]
    L.oracle_dbscan_batch.restype = ctypes.c_int


def _p(a, t):
    return a.ctypes.data_as(t)


def dbscan(xy, eps, minpts, kdapi=False, with_members=False):
    """One DBSCAN::Run().  Returns (rc, labels[int32 n], n_clusters[, clusters list])."""
    xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
    n = xy.shape[0]
    labels = np.full(max(n, 1), -1, dtype=np.int32)
    nc = ctypes.c_uint32(0)
    members = np.zeros(max(n, 1), dtype=np.uint32)
    moff = np.zeros(n + 2, dtype=np.uint32)
    args = [_p(xy, _dp), n, float(eps), int(minpts), _p(labels, _i32p), ctypes.byref(nc), _p(members, _u32p),
            _p(moff, _u32p)]
    if kdapi:
        rc = lib().oracle_dbscan_kdapi(REF_KDTREE_SO.encode(), *args)
    else:
        rc = lib().oracle_dbscan(*args)
    labels = labels[:n]
    if with_members:
        cl = [members[moff[c]:moff[c + 1]].copy() for c in range(nc.value)]
        return rc, labels, nc.value, cl
    return rc, labels, nc.value


def range_query_all(xy, eps, kdapi=False):
    """CSR (off, idx) of every point's raw range query (self included, list order)."""
    xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
    n = xy.shape[0]
    cap = max(16, n * 64)
    while True:
        off = np.zeros(n + 1, dtype=np.uint64)
        idx = np.zeros(cap, dtype=np.uint32)
        if kdapi:
            r = lib().oracle_range_query_all_kdapi(REF_KDTREE_SO.encode(), _p(xy, _dp), n, float(eps),
                                                   _p(off, _u64p), _p(idx, _u32p), cap)
        else:
            r = lib().oracle_range_query_all(_p(xy, _dp), n, float(eps), _p(off, _u64p), _p(idx, _u32p), cap)
        if r == -2:
            cap *= 4
            continue
        if r < 0:
            raise RuntimeError("oracle range query failed: %d" % r)
        return off, idx[:r]


def dbscan_batch(xy, seg_off, seg_cnt, eps, minpts):
    xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
    seg_off = np.ascontiguousarray(seg_off, dtype=np.uint32)
    seg_cnt = np.ascontiguousarray(seg_cnt, dtype=np.uint32)
    S = seg_off.shape[0]
    labels = np.full(max(xy.shape[0], 1), -1, dtype=np.int32)
    ncl = np.zeros(max(S, 1), dtype=np.uint32)
    lib().oracle_dbscan_batch(_p(xy, _dp), _p(seg_off, _u32p), _p(seg_cnt, _u32p), S, float(eps), int(minpts),
                              _p(labels, _i32p), _p(ncl, _u32p))
    return labels[:xy.shape[0]], ncl[:S]


# ---- ingest + slicing (oracle/event_oracle.cpp) ----
_u8p = ctypes.POINTER(ctypes.c_uint8)


def _declare_events(L):
    L.oracle_window_bounds.argtypes = [_u8p, ctypes.c_uint64, ctypes.c_double, ctypes.c_double, _u64p, _u64p]
    L.oracle_window_bounds.restype = ctypes.c_int
    L.oracle_check_sorted.argtypes = [_u8p, ctypes.c_uint64]
    L.oracle_check_sorted.restype = ctypes.c_int
    L.oracle_event_frame.argtypes = [_u8p, ctypes.c_uint64, ctypes.c_uint64, _dp, _u32p, _u32p, _i32p]
    L.oracle_event_frame.restype = ctypes.c_int
    L.oracle_event_frame_ref.argtypes = L.oracle_event_frame.argtypes
    L.oracle_event_frame_ref.restype = ctypes.c_int
    L.oracle_event_frame_model.argtypes = [_u8p, ctypes.c_uint64, ctypes.c_uint64, _u64p, ctypes.c_uint32, _dp, _u32p, _u32p,
                                           _i32p]
    L.oracle_event_frame_model.restype = ctypes.c_int
    L.oracle_pixel_hash.argtypes = [ctypes.c_double, ctypes.c_double]
    L.oracle_pixel_hash.restype = ctypes.c_uint64
    L.oracle_pixel_hash_restated.argtypes = [ctypes.c_double, ctypes.c_double]
    L.oracle_pixel_hash_restated.restype = ctypes.c_uint64
    L.oracle_bucket_counts.argtypes = [ctypes.c_uint32, _u64p]
    L.oracle_bucket_counts.restype = None
    L.oracle_next_bkt.argtypes = [ctypes.c_uint64]
    L.oracle_next_bkt.restype = ctypes.c_uint64


def window_bounds(rec, t0, t1):
    L = lib()
    _declare_events(L)
    rec = np.ascontiguousarray(rec, dtype=np.uint8)
    lo, hi = ctypes.c_uint64(0), ctypes.c_uint64(0)
    L.oracle_window_bounds(_p(rec, _u8p), rec.size // 25, float(t0), float(t1), ctypes.byref(lo), ctypes.byref(hi))
    return lo.value, hi.value


def check_sorted(rec):
    L = lib()
    _declare_events(L)
    rec = np.ascontiguousarray(rec, dtype=np.uint8)
    return L.oracle_check_sorted(_p(rec, _u8p), rec.size // 25)


def bucket_steps(n_steps=28):
    """libstdc++'s unordered_set bucket counts, epoch by epoch (13, 29, 59, ...), from the library's own policy object."""
    L = lib()
    _declare_events(L)
    out = [int(L.oracle_next_bkt(12))]
    while len(out) < n_steps:
        out.append(int(L.oracle_next_bkt(2 * out[-1])))
    return np.array(out, dtype=np.uint64)


def event_frame(rec, lo, hi, order="canonical"):
    """EventFrame ctor on events [lo,hi): (xy_pos [nP,2], xy_neg [nN,2], event_point int32 [hi-lo]).
    order: "canonical" (first occurrence), "reference" (real std::unordered_set, EventFrame.cpp:34-35) or "model"
    (the restated list rules the HIP slicer follows)."""
    L = lib()
    _declare_events(L)
    rec = np.ascontiguousarray(rec, dtype=np.uint8)
    n = hi - lo
    xy = np.zeros((max(n, 1), 2), dtype=np.float64)
    ep = np.full(max(n, 1), -1, dtype=np.int32)
    npos, nneg = ctypes.c_uint32(0), ctypes.c_uint32(0)
    if order == "model":
        st = bucket_steps()
        rc = L.oracle_event_frame_model(_p(rec, _u8p), lo, hi, _p(st, _u64p), len(st), _p(xy, _dp), ctypes.byref(npos),
                                        ctypes.byref(nneg), _p(ep, _i32p))
        assert rc == 0
    else:
        fn = {"canonical": L.oracle_event_frame, "reference": L.oracle_event_frame_ref}[order]
        fn(_p(rec, _u8p), lo, hi, _p(xy, _dp), ctypes.byref(npos), ctypes.byref(nneg), _p(ep, _i32p))
    return xy[:npos.value].copy(), xy[npos.value:npos.value + nneg.value].copy(), ep[:n]


def pack_events(t, x, y, p):
    """numpy arrays -> packed 25-byte records (uint8 [n*25])."""
    t = np.asarray(t, np.float64)
    n = t.shape[0]
    rec = np.zeros((n, 25), dtype=np.uint8)
    rec[:, 0:8] = t.view(np.uint8).reshape(n, 8)
    rec[:, 8:16] = np.asarray(x, np.float64).view(np.uint8).reshape(n, 8)
    rec[:, 16:24] = np.asarray(y, np.float64).view(np.uint8).reshape(n, 8)
    rec[:, 24] = np.asarray(p, np.uint8)
    return rec.reshape(-1)


def detect_windows(rec, t0, t1, eps, minpts, cluster_min=5, need_clusters=36, radius_thr=15.511363636363637):
    """CPU baseline loop (oracle/pipeline_oracle.cpp): returns (events covered, total kept clusters)."""
    L = lib()
    L.oracle_detect_windows.argtypes = [_u8p, ctypes.c_uint64, _dp, _dp, ctypes.c_uint32, ctypes.c_double,
                                        ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_double, _u64p]
    L.oracle_detect_windows.restype = ctypes.c_uint64
    rec = np.ascontiguousarray(rec, dtype=np.uint8)
    t0 = np.ascontiguousarray(t0, dtype=np.float64)
    t1 = np.ascontiguousarray(t1, dtype=np.float64)
    ncl = ctypes.c_uint64(0)
    ev = L.oracle_detect_windows(_p(rec, _u8p), rec.size // 25, _p(t0, _dp), _p(t1, _dp), t0.shape[0], float(eps),
                                 int(minpts), int(cluster_min), int(need_clusters), float(radius_thr),
                                 ctypes.byref(ncl))
    return int(ev), int(ncl.value)


def detect_windows_mt(rec, t0, t1, eps, minpts, n_threads, cluster_min=5, need_clusters=36, radius_thr=15.511363636363637):
    """The same loop on n_threads workers over 5 n_threads pieces (the reference driver's threading)."""
    L = lib()
    L.oracle_detect_windows_mt.argtypes = [_u8p, ctypes.c_uint64, _dp, _dp, ctypes.c_uint32, ctypes.c_double, ctypes.c_uint32,
                                           ctypes.c_uint32, ctypes.c_uint32, ctypes.c_double, ctypes.c_uint32, _u64p]
    L.oracle_detect_windows_mt.restype = ctypes.c_uint64
    rec = np.ascontiguousarray(rec, dtype=np.uint8)
    t0 = np.ascontiguousarray(t0, dtype=np.float64)
    t1 = np.ascontiguousarray(t1, dtype=np.float64)
    ncl = ctypes.c_uint64(0)
    ev = L.oracle_detect_windows_mt(_p(rec, _u8p), rec.size // 25, _p(t0, _dp), _p(t1, _dp), t0.shape[0], float(eps), int(minpts),
                                    int(cluster_min), int(need_clusters), float(radius_thr), int(n_threads), ctypes.byref(ncl))
    return int(ev), int(ncl.value)


def set_kd_backend(reference):
    """Which k-d tree runs under oracle_dbscan and everything built on it (extract_candidates, the window loops):
    True -> the reference's own kdtree.cpp as compiled into oracle/_ref/libkdtree_ref.so (kept mapped), False -> the
    restated tree of oracle/dbscan_oracle.cpp.  Raises when the reference build is asked for and absent."""
    L = lib()
    L.oracle_set_kd_backend.argtypes = [ctypes.c_char_p]
    L.oracle_set_kd_backend.restype = ctypes.c_int
    if reference:
        if not have_ref_kdtree() or L.oracle_set_kd_backend(REF_KDTREE_SO.encode()) != 0:
            raise RuntimeError("oracle/_ref/libkdtree_ref.so (the reference's kdtree.cpp, built by `make -C oracle`) is not available")
    else:
        L.oracle_set_kd_backend(None)


def kd_backend():
    """'reference kdtree.cpp (oracle/_ref)' or 'restated kd-tree (oracle/dbscan_oracle.cpp)': what oracle_dbscan runs on."""
    L = lib()
    L.oracle_kd_backend.restype = ctypes.c_int
    return "reference kdtree.cpp (oracle/_ref)" if L.oracle_kd_backend() else "restated kd-tree (oracle/dbscan_oracle.cpp)"


def detect_windows_full(rec, t0, t1, win_base, slots, eps=4.0, minpts=2, cluster_min=5, need_clusters=36,
                        radius_thr=15.511363636363637, fit_circle=False, knn_num=3, n_threads=None):
    """Every result of every window in the device pipeline's slot layout (oracle_detect_windows_full_mt): a dict of numpy
    arrays win_lo, win_hi, seg_cnt [2S], n_clusters [2S], win_info [S,4], tie [S], xy [slots,2], event_point, labels,
    kept_labels, rep [slots], cand_pair [slots,2], cand_xyr [slots,3] and the byte masks def_pts / def_kept / def_rep /
    def_cand [slots] of the slots the reference defines.  win_base: S + 1 slot offsets (exclusive scan of window sizes)."""
    L = lib()
    _u8 = _u8p
    L.oracle_detect_windows_full_mt.argtypes = [_u8, ctypes.c_uint64, _dp, _dp, ctypes.c_uint32, ctypes.c_double, ctypes.c_uint32,
                                                ctypes.c_uint32, ctypes.c_uint32, ctypes.c_double, ctypes.c_int, ctypes.c_uint32,
                                                ctypes.c_uint32, _u64p, _u64p, _u64p, _u32p, _u32p, _u32p, _u8, _dp, _i32p, _i32p,
                                                _i32p, _u32p, _u32p, _dp, _u8, _u8, _u8, _u8]
    L.oracle_detect_windows_full_mt.restype = ctypes.c_uint64
    rec = np.ascontiguousarray(rec, dtype=np.uint8)
    t0 = np.ascontiguousarray(t0, dtype=np.float64)
    t1 = np.ascontiguousarray(t1, dtype=np.float64)
    S = t0.shape[0]
    wb = np.ascontiguousarray(win_base, dtype=np.uint64)
    assert wb.shape[0] == S + 1 and int(wb[-1]) <= slots
    if n_threads is None:
        n_threads = max(1, (os.cpu_count() or 1) - 2)
    sl = max(int(slots), 1)
    out = dict(win_lo=np.zeros(S, np.uint64), win_hi=np.zeros(S, np.uint64), seg_cnt=np.zeros(2 * S, np.uint32),
               n_clusters=np.zeros(2 * S, np.uint32), win_info=np.zeros((S, 4), np.uint32), tie=np.zeros(S, np.uint8),
               xy=np.zeros((sl, 2)), event_point=np.zeros(sl, np.int32), labels=np.zeros(sl, np.int32),
               kept_labels=np.zeros(sl, np.int32), rep=np.zeros(sl, np.uint32), cand_pair=np.zeros((sl, 2), np.uint32),
               cand_xyr=np.zeros((sl, 3)), def_pts=np.zeros(sl, np.uint8), def_kept=np.zeros(sl, np.uint8),
               def_rep=np.zeros(sl, np.uint8), def_cand=np.zeros(sl, np.uint8))
    o = out
    ev = L.oracle_detect_windows_full_mt(_p(rec, _u8), rec.size // 25, _p(t0, _dp), _p(t1, _dp), S, float(eps), int(minpts),
                                         int(cluster_min), int(need_clusters), float(radius_thr), int(bool(fit_circle)), int(knn_num),
                                         int(n_threads), _p(wb, _u64p), _p(o["win_lo"], _u64p), _p(o["win_hi"], _u64p),
                                         _p(o["seg_cnt"], _u32p), _p(o["n_clusters"], _u32p), _p(o["win_info"], _u32p),
                                         _p(o["tie"], _u8), _p(o["xy"], _dp), _p(o["event_point"], _i32p), _p(o["labels"], _i32p),
                                         _p(o["kept_labels"], _i32p), _p(o["rep"], _u32p), _p(o["cand_pair"], _u32p),
                                         _p(o["cand_xyr"], _dp), _p(o["def_pts"], _u8), _p(o["def_kept"], _u8), _p(o["def_rep"], _u8),
                                         _p(o["def_cand"], _u8))
    if ev == (1 << 64) - 1:
        raise RuntimeError("a window does not fit its slots (win_base is not the scan of the windows' sizes)")
    out["events"] = int(ev)
    out["n_threads"] = int(n_threads)
    return out


# ---- circle-candidate extraction (oracle/detect_oracle.cpp) ----
def circle_radius_threshold(width, height, rows, cols, asymmetric, square, radius):
    L = lib()
    L.oracle_circle_radius_threshold.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_int, ctypes.c_double, ctypes.c_double]
    L.oracle_circle_radius_threshold.restype = ctypes.c_double
    return L.oracle_circle_radius_threshold(width, height, rows, cols, int(asymmetric), square, radius)


def extract_candidates(pos, neg, eps, minpts, cluster_min, need_clusters, radius_thr, fit_circle=False, knn_num=3,
                       override_pos=None, override_neg=None):
    """extractFeatures up to the candidate list (both fitCircle paths).  Returns a dict; tie_pos / tie_neg flag the kept
    clusters whose nth_element median has an equal-norm rival (the reference's choice then depends on its BFS member
    order); override_pos / override_neg (uint32 per kept cluster, 0xFFFFFFFF = keep) hand in other representatives."""
    L = lib()
    L.oracle_extract_candidates_override.argtypes = [_dp, ctypes.c_uint32, _dp, ctypes.c_uint32, ctypes.c_double,
                                                     ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_double,
                                                     ctypes.c_int, ctypes.c_uint32, _u32p, _u32p, _dp, _i32p, _i32p,
                                                     _u32p, _u32p, _u32p, _u32p, _u32p, _u32p]
    L.oracle_extract_candidates_override.restype = ctypes.c_int
    pos = np.ascontiguousarray(pos, np.float64).reshape(-1, 2)
    neg = np.ascontiguousarray(neg, np.float64).reshape(-1, 2)
    npos, nneg = pos.shape[0], neg.shape[0]
    info = np.zeros(4, np.uint32)
    m = max(npos, nneg, 1)
    pair = np.zeros((m, 2), np.uint32)
    xyr = np.zeros((m, 3), np.float64)
    kp = np.full(max(npos, 1), -1, np.int32)
    kn = np.full(max(nneg, 1), -1, np.int32)
    rp = np.zeros(max(npos, 1), np.uint32)
    rn = np.zeros(max(nneg, 1), np.uint32)
    tp = np.zeros(max(npos, 1), np.uint32)
    tn = np.zeros(max(nneg, 1), np.uint32)
    op = None if override_pos is None else np.ascontiguousarray(override_pos, np.uint32)
    on = None if override_neg is None else np.ascontiguousarray(override_neg, np.uint32)
    L.oracle_extract_candidates_override(_p(pos, _dp), npos, _p(neg, _dp), nneg, float(eps), int(minpts), int(cluster_min),
                                         int(need_clusters), float(radius_thr), int(bool(fit_circle)), int(knn_num),
                                         _p(info, _u32p), _p(pair, _u32p), _p(xyr, _dp), _p(kp, _i32p), _p(kn, _i32p),
                                         _p(rp, _u32p), _p(rn, _u32p), _p(tp, _u32p), _p(tn, _u32p),
                                         None if op is None else _p(op, _u32p), None if on is None else _p(on, _u32p))
    nc = int(info[0])
    return dict(n=nc, nk_pos=int(info[1]), nk_neg=int(info[2]), status=int(info[3]) & 1, tie=bool(info[3] & 2),
                pair=pair[:nc], xyr=xyr[:nc], kept_pos=kp[:npos], kept_neg=kn[:nneg], rep_pos=rp[:int(info[1])],
                rep_neg=rn[:int(info[2])], tie_pos=tp[:int(info[1])].astype(bool), tie_neg=tn[:int(info[2])].astype(bool))


def fit_circle(a, b):
    L = lib()
    L.oracle_fit_circle.argtypes = [_dp, ctypes.c_uint32, _dp, ctypes.c_uint32, _dp, _dp]
    a = np.ascontiguousarray(a, np.float64).reshape(-1, 2)
    b = np.ascontiguousarray(b, np.float64).reshape(-1, 2)
    c = np.zeros(2)
    r = ctypes.c_double(0)
    L.oracle_fit_circle(_p(a, _dp), a.shape[0], _p(b, _dp), b.shape[0], _p(c, _dp), ctypes.byref(r))
    return c, r.value


# ---- solver oracle (oracle/solver_oracle.cpp) ----
def solver_evaluate(problem, x, want_H=True):
    """Dense CPU evaluation of one-segment problems: (cost, g, H) in tangent order [intr | cp (rot3, trans3)]."""
    L = lib()
    L.oracle_evaluate_mode.argtypes = [_dp, ctypes.c_uint32, _dp, _dp, _dp, ctypes.c_uint64, _dp, _dp, _u32p, _dp,
                                       ctypes.c_double, ctypes.c_double, ctypes.c_int, _dp, _dp]
    L.oracle_evaluate_mode.restype = ctypes.c_double
    n_cp = int(problem["seg_cp_off"][-1])
    assert len(problem["seg_cp_off"]) == 2, "oracle_evaluate handles one segment"
    x = np.ascontiguousarray(x, np.float64)
    intr = x[:9].copy()
    q = x[9:9 + 4 * n_cp].copy()
    t = x[9 + 4 * n_cp:].copy()
    kn = np.ascontiguousarray(problem["knots"], np.float64)
    obs = np.ascontiguousarray(problem["obs"], np.float64)
    tm = np.ascontiguousarray(problem["time"], np.float64)
    lm = np.ascontiguousarray(problem["lm_id"], np.uint32)
    lms = np.ascontiguousarray(problem["landmarks"], np.float64)
    n = 9 + 6 * n_cp
    g = np.zeros(n)
    H = np.zeros((n, n)) if want_H else None
    cost = L.oracle_evaluate_mode(_p(intr, _dp), n_cp, _p(q, _dp), _p(t, _dp), _p(kn, _dp), tm.shape[0], _p(obs, _dp),
                                  _p(tm, _dp), _p(lm, _u32p), _p(lms, _dp), float(problem["circle_radius"]),
                                  float(problem["huber_a"]),
                                  int(bool(problem.get("use_so3", False))) | (2 if problem.get("fisheye", False) else 0), _p(g, _dp),
                                  _p(H, _dp) if want_H else None)
    return cost, g, H


def solver_evaluate_arrow(problem, x, n_threads=1):
    """The normal equations of a one-segment problem of any size in the product's accumulation-buffer layout
    (oracle_evaluate_arrow_mt: [0] cost | g_intr 9 | H_intr 81 (upper) | per control point 204: g 6, H_c,intr 54, H_c,c+d 4 x 36),
    on n_threads host threads."""
    L = lib()
    L.oracle_evaluate_arrow_mt.argtypes = [_dp, ctypes.c_uint32, _dp, _dp, _dp, ctypes.c_uint64, _dp, _dp, _u32p, _dp,
                                           ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int, _dp]
    L.oracle_evaluate_arrow_mt.restype = None
    n_cp = int(problem["seg_cp_off"][-1])
    assert len(problem["seg_cp_off"]) == 2, "one segment"
    x = np.ascontiguousarray(x, np.float64)
    intr = x[:9].copy()
    q = x[9:9 + 4 * n_cp].copy()
    t = x[9 + 4 * n_cp:].copy()
    kn = np.ascontiguousarray(problem["knots"], np.float64)
    obs = np.ascontiguousarray(problem["obs"], np.float64)
    tm = np.ascontiguousarray(problem["time"], np.float64)
    lm = np.ascontiguousarray(problem["lm_id"], np.uint32)
    lms = np.ascontiguousarray(problem["landmarks"], np.float64)
    acc = np.zeros(91 + 204 * n_cp)
    L.oracle_evaluate_arrow_mt(_p(intr, _dp), n_cp, _p(q, _dp), _p(t, _dp), _p(kn, _dp), tm.shape[0], _p(obs, _dp), _p(tm, _dp),
                               _p(lm, _u32p), _p(lms, _dp), float(problem["circle_radius"]), float(problem["huber_a"]),
                               int(bool(problem.get("use_so3", False))) | (2 if problem.get("fisheye", False) else 0), int(n_threads),
                               _p(acc, _dp))
    return acc


def inverse_radial(k4):
    L = lib()
    L.oracle_inverse_radial.argtypes = [_dp, _dp]
    k = np.ascontiguousarray(k4, np.float64)
    b = np.zeros(5)
    L.oracle_inverse_radial(_p(k, _dp), _p(b, _dp))
    return b


def associate(rec, kf_time, circles, t_min, t_max, max_dt, edge_tol):
    """EventCalibSpline association: returns (obs [m,2], time [m], lm_id [m])."""
    L = lib()
    L.oracle_associate.argtypes = [_u8p, ctypes.c_uint64, _dp, _dp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_double,
                                   ctypes.c_double, ctypes.c_double, ctypes.c_double, _dp, _dp, _u32p]
    L.oracle_associate.restype = ctypes.c_uint64
    rec = np.ascontiguousarray(rec, np.uint8)
    n = rec.size // 25
    kf = np.ascontiguousarray(kf_time, np.float64)
    ci = np.ascontiguousarray(circles, np.float64)
    K = kf.shape[0]
    nc = ci.reshape(K, -1, 3).shape[1] if K else 0
    obs = np.zeros((max(n, 1), 2))
    tm = np.zeros(max(n, 1))
    lm = np.zeros(max(n, 1), np.uint32)
    m = L.oracle_associate(_p(rec, _u8p), n, _p(kf, _dp), _p(ci, _dp), K, nc, float(t_min), float(t_max), float(max_dt),
                           float(edge_tol), _p(obs, _dp), _p(tm, _dp), _p(lm, _u32p))
    return obs[:m].copy(), tm[:m].copy(), lm[:m].copy()


def rectify(pos, neg, kept_pos, kept_neg, pose, camera, dist, width, height, landmarks, rows, cols, asymmetric,
            circle_radius, fit_circle=False, model=0):
    """CirclesEventFrame::rectifyFeatures for one keyframe -> (feat_xyr [n,3], valid [n], ok, erased).  model 1: fisheye
    projections (dist[0..3] = k1..k4)."""
    L = lib()
    L.oracle_rectify_cam.restype = ctypes.c_int
    pos = np.ascontiguousarray(pos, np.float64).reshape(-1, 2)
    neg = np.ascontiguousarray(neg, np.float64).reshape(-1, 2)
    kp = np.ascontiguousarray(kept_pos, np.int32)
    kn = np.ascontiguousarray(kept_neg, np.int32)
    pose = np.ascontiguousarray(pose, np.float64).reshape(12)
    camera = np.ascontiguousarray(camera, np.float64).reshape(4)
    dist = np.ascontiguousarray(dist, np.float64).reshape(5)
    lm = np.ascontiguousarray(landmarks, np.float64).reshape(-1, 3)
    n = rows * cols
    feat = np.zeros((n, 3), np.float64)
    valid = np.zeros(n, np.uint32)
    info = np.zeros(2, np.uint32)
    vp = ctypes.c_void_p
    L.oracle_rectify_cam(vp(pos.ctypes.data), ctypes.c_uint32(len(pos)), vp(neg.ctypes.data), ctypes.c_uint32(len(neg)),
                     vp(kp.ctypes.data), vp(kn.ctypes.data), vp(pose.ctypes.data), vp(camera.ctypes.data),
                     vp(dist.ctypes.data), ctypes.c_double(width), ctypes.c_double(height), vp(lm.ctypes.data),
                     ctypes.c_uint32(rows), ctypes.c_uint32(cols), ctypes.c_int(int(asymmetric)),
                     ctypes.c_double(circle_radius), ctypes.c_int(int(fit_circle)), vp(feat.ctypes.data),
                     vp(valid.ctypes.data), vp(info.ctypes.data), ctypes.c_int(int(model)))
    return feat, valid, int(info[0]), int(info[1])


# ---- adaptive windowing + keyframe gate (oracle/policy_oracle.cpp) ----
_DETECT_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.POINTER(ctypes.c_int),
                              ctypes.POINTER(ctypes.c_double))


def policy_run(detect, start_time, end_time, piece_num, motion_time_step, frame_event_num_threshold, rows=9, cols=4, mode=0,
               max_keyframes=1 << 16):
    """MultiProcess::process + EventCalibIni::track on the CPU.  detect(t0, t1) -> (found, events_num, features [rows*cols, 3]
    or None) stands for CirclesEventFrame(...).extractFeatures() on the window.  mode 0: the build's own-piece gate, 1: the
    reference's single shared map with one worker.  Returns dict(time, duration, events_num, features, windows)."""
    L = lib()
    L.oracle_policy_run.argtypes = [_DETECT_FN, ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_double,
                                    ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, _dp, _dp, _i32p, _dp,
                                    _u64p]
    L.oracle_policy_run.restype = ctypes.c_int64
    M = rows * cols
    err = []

    def cb(user, t0, t1, ev_out, feat_out):
        try:
            found, events_num, feat = detect(t0, t1)
            ev_out[0] = int(events_num)
            if found:
                f = np.ascontiguousarray(feat, np.float64).reshape(M, 3)
                ctypes.memmove(feat_out, f.ctypes.data, 8 * 3 * M)
            return 1 if found else 0
        except Exception as e:      # never let an exception cross the C frame
            err.append(e)
            ev_out[0] = 0
            return 0
    t = np.zeros(max_keyframes)
    d = np.zeros((max_keyframes, 2))
    e = np.zeros(max_keyframes, np.int32)
    f = np.zeros((max_keyframes, M, 3))
    w = ctypes.c_uint64(0)
    K = L.oracle_policy_run(_DETECT_FN(cb), None, float(start_time), float(end_time), int(piece_num), float(motion_time_step),
                            int(frame_event_num_threshold), rows, cols, int(mode), max_keyframes, _p(t, _dp), _p(d, _dp),
                            _p(e, _i32p), _p(f, _dp), ctypes.byref(w))
    if err:
        raise err[0]
    assert 0 <= K <= max_keyframes, K
    return dict(time=t[:K].copy(), duration=d[:K].copy(), events_num=e[:K].astype(np.int64), features=f[:K].copy(),
                windows=int(w.value))


def track_gate(ref_feat, ref_time, cur_feat, cur_time, rows, cols, motion_time_step):
    L = lib()
    L.oracle_track_gate.argtypes = [_dp, ctypes.c_double, _dp, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_double]
    L.oracle_track_gate.restype = ctypes.c_int
    a = np.ascontiguousarray(ref_feat, np.float64).reshape(rows * cols, 3)
    b = np.ascontiguousarray(cur_feat, np.float64).reshape(rows * cols, 3)
    return bool(L.oracle_track_gate(_p(a, _dp), float(ref_time), _p(b, _dp), float(cur_time), rows, cols, float(motion_time_step)))
