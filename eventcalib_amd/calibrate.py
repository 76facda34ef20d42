"""The reference driver end to end over a device-resident event stream.

Python mirror of unit_test_eventCameraCalib's main (event_camera_calib/test/eventCameraCalib.cpp:99-233) and of the
C++ shims in eventcalib_amd/csrc/host/ (multi_process.hpp, event_calib_ini.hpp, event_calib_spline.hpp):

  1. keyframe search: adaptive windows + grid ordering + orientation gate          eventCameraCalib.cpp:168-190
  2. EventCalibIni::cvCalibration: frame selection, calibrateCamera, solvePnPRansac for every keyframe, checkPose,
     rectifyFeatures                                                              EventCalibIni.cpp:149-325
  3. EventCalibSpline: gap segmentation, spline initialisation, event association, the continuous-time solve,
     updateMap                                                                    EventCalibSpline.cpp:14-317
  4. SystemBase::saveKeyFrameTrajectoryTUM                                        SystemBase.cpp:122-150

Every numeric stage is a libecal.so entry point (HIP kernels or, for the two tiny banded fits, host C++); this
module is orchestration only — torch owns HBM buffers, numpy carries the small per-keyframe tables.
"""
import numpy as np
import torch

from . import capi
from .adaptive import detect_keyframes_device
from .pipeline import DetectPipeline

EXAMPLE_FLAGS = (capi.CALIB_FIX_ASPECT_RATIO | capi.CALIB_FIX_PRINCIPAL_POINT | capi.CALIB_ZERO_TANGENT_DIST |
                 capi.CALIB_FIX_K4 | capi.CALIB_FIX_K5 | capi.CALIB_FIX_K6)       # parameters.hpp:47-58 on example.yaml
# Calibrate_UseFisheyeModel: 1 — the fisheye enum overwrites the flags (parameters.hpp:59-68) — with Fix_K2 .. Fix_K4: 1: from
# circle centres good to ~3 px a 55-degree lens shows k1 only; free, k2 and k3 come out in the tens (the fit follows the noise at
# the image border) and their series reversion starts the spline solve in the basin of a wrong minimum (measured: principal point
# 4 px off, fx 1 % off).  The spline solve itself frees all five inverse coefficients, as the reference does for the radial model.
EXAMPLE_FLAGS_FISHEYE = (capi.CALIB_FIX_SKEW | capi.CALIB_RECOMPUTE_EXTRINSIC | capi.CALIB_FIX_PRINCIPAL_POINT |
                         capi.CALIB_FIX_K2 | capi.CALIB_FIX_K3 | capi.CALIB_FIX_K4)


def board_points(rows=9, cols=4, square=5.5, asymmetric=True):
    """EventCalibIni::calcBoardCornerPositions (EventCalibIni.cpp:99-115), narrowed to float like cv::Point3f."""
    pts = [(((2 * j + i % 2) if asymmetric else j) * square, i * square, 0.0) for i in range(rows) for j in range(cols)]
    return np.array(pts, np.float32).astype(np.float64)


def rodrigues(rv):
    """[n,3] rotation vectors -> [n,3,3] (cv::Rodrigues)."""
    rv = np.asarray(rv, np.float64).reshape(-1, 3)
    th = np.linalg.norm(rv, axis=1)
    k = rv / np.maximum(th, 1e-300)[:, None]
    K = np.zeros((len(rv), 3, 3))
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -k[:, 2], k[:, 1], k[:, 2], -k[:, 0], -k[:, 1], k[:, 0]
    return (np.cos(th)[:, None, None] * np.eye(3) + (1 - np.cos(th))[:, None, None] * k[:, :, None] * k[:, None, :] +
            np.sin(th)[:, None, None] * K)


def quat_from_matrix(R):
    """[n,3,3] -> [n,4] x y z w (Eigen::Quaterniond(Matrix3d)'s branches)."""
    R = np.asarray(R, np.float64)
    out = np.zeros((len(R), 4))
    tr = R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2]
    pos = tr > 0
    if pos.any():   # the usual branch, all such matrices at once
        m = R[pos]
        t = np.sqrt(tr[pos] + 1.0)
        w = 0.5 * t
        t = 0.5 / t
        out[pos] = np.stack([(m[:, 2, 1] - m[:, 1, 2]) * t, (m[:, 0, 2] - m[:, 2, 0]) * t, (m[:, 1, 0] - m[:, 0, 1]) * t, w], axis=1)
    for n in np.flatnonzero(~pos):
        m = R[n]
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0)
        out[n, i] = 0.5 * t
        t = 0.5 / t
        out[n, 3] = (m[k, j] - m[j, k]) * t
        out[n, j] = (m[j, i] + m[i, j]) * t
        out[n, k] = (m[k, i] + m[i, k]) * t
    return out


def check_pose(R_ref, twb_ref, t_ref, R_cur, twb_cur, t_cur, step):
    """EventCalibIni::checkPose (EventCalibIni.cpp:327-347)."""
    import math
    dt = float(t_cur - t_ref)
    d = [float(twb_cur[i]) - float(twb_ref[i]) for i in range(3)]
    v_t = math.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) / dt
    a, b = np.asarray(R_cur, np.float64).ravel().tolist(), np.asarray(R_ref, np.float64).ravel().tolist()
    tr = 0.0                                     # trace(R_cur R_ref^T) = sum of the elementwise products
    for i in range(9):
        tr += a[i] * b[i]
    c = (tr - 1) * 0.5
    v_r = abs(math.acos(min(1.0, max(-1.0, c))) / dt)
    return v_t < (2.5e-1 / step) * 2 and v_r < (5e-4 * math.pi) * 2 / step


def calibrate_stream(ctx, events, t_first, t_last, *, motion_time_step=5e-4, frame_event_num_threshold=4000, piece_num=30,
                     frames_to_use=200, width=346.0, height=260.0, rows=9, cols=4, square=5.5, circle_radius=1.75,
                     flags=None, aspect_ratio=1.0, use_so3=False, max_num_iterations=50, eps=4.0, minpts=2,
                     gate_mode=capi.GATE_SHARED_MAP, fisheye=False, tables=False):
    """events: uint8 CUDA tensor of packed 25-byte records.  Returns a dict with the initial calibration, the refined
    intrinsics [fx fy cx cy k1..k5 (inverse radial polynomial)] and the keyframe trajectory.
    gate_mode: capi.GATE_SHARED_MAP (default: the reference's keyframe gate as its single-worker run computes it — one keyframe
    map for all pieces) or capi.GATE_OWN_PIECE (the schedule-free own-piece gate: 2 - 3 x faster keyframe search, every piece's
    first success ungated).
    fisheye (Calibrate_UseFisheyeModel: 1; BASELINE configs[4]): cv::fisheye::calibrate's model in the init stage
    (EventCalibIni.cpp:186-190), and — new: the reference stops at EventCalibSpline.cpp:97-99 — the Kannala-Brandt camera in
    the PnP, in rectifyFeatures' projections and in the spline solve (k1..k5 = the inverse angle polynomial).
    The detection pipeline of the rectification stage stays parked on the context between calls (`ctx._calibrate_pipe`: arrays
    sized for the stream, ~2.5 GB of HBM for 50 M events); `release_scratch(ctx)` frees it when a long-lived context will not
    calibrate again.
    tables: also return the per-stage tables (keyframe records, PnP poses and verdicts, rectified circles, accepted frames, the
    spline's start) that tests/test_gpu_oracle_chain.py compares with the CPU oracle chain."""
    if flags is None:
        flags = EXAMPLE_FLAGS_FISHEYE if fisheye else EXAMPLE_FLAGS
    model = 1 if fisheye else 0
    import time as _time
    dev = events.device
    st = torch.cuda.current_stream(dev).cuda_stream
    step = motion_time_step
    n_circ = rows * cols
    stages = {}
    t_mark = [_time.perf_counter()]

    def mark(name):   # wall seconds per stage (the GPU is drained at every mark: stages do not overlap)
        torch.cuda.synchronize(dev)
        now = _time.perf_counter()
        stages[name] = stages.get(name, 0.0) + now - t_mark[0]
        t_mark[0] = now
    # (the detection pipeline of the rectification stage is kept with the context: its arrays are sized for the stream — 2.5 GB
    # for 50 M events —, and getting them from the runtime anew cost every call 7 ms on top of the stage's 3)
    pipe = getattr(ctx, "_calibrate_pipe", None)
    if pipe is None or pipe.dev != dev:
        pipe = DetectPipeline(ctx, dev)
        try:
            ctx._calibrate_pipe = pipe
        except AttributeError:
            pass
    # -- 1. keyframes
    kf = detect_keyframes_device(pipe.ctx, events, step, frame_event_num_threshold, piece_num, t_first, t_last, eps, minpts, rows, cols,
                                 gate_mode=gate_mode)
    K = len(kf["time"])
    out = {"keyframes": K, "stage_seconds": stages}
    mark("keyframe_search")
    if K == 0:
        raise RuntimeError("no keyframe found")
    # -- 2. init calibration on a subset (EventCalibIni.cpp:163-181), image points narrowed to float like cv::Point2f
    use = frames_to_use
    sel_step = K // use
    if sel_step == 0:
        use, sel_step = K, 1
    sel = np.arange(use) * sel_step
    obj = board_points(rows, cols, square)
    feat32 = kf["features"][:, :, :2].astype(np.float32).astype(np.float64)
    if fisheye:
        # the library's one start procedure (ecal_calibrate_fisheye_views, the same the C++ shim calls): cv::fisheye::calibrate's
        # own start first (f = max(w, h) / pi), the radial model's focal lengths as a guess only when that fails
        ini = capi.calibrate_fisheye_views(ctx, obj, feat32[sel], width, height, flags, aspect_ratio)
        out["fisheye_start"] = {0: "reference (f = max(w, h) / pi)", 1: "radial model's focal lengths (the reference's start failed)",
                                2: "caller's guess"}[ini["start_used"]]
    else:
        ini = capi.calibrate_views(ctx, obj, feat32[sel], width, height, model, flags, aspect_ratio)
    intr0 = ini["intr"]
    out["init"] = {"intr": intr0, "rms": ini["rms"], "iterations": ini["iterations"], "views": use}
    mark("init_calibration")
    # solvePnPRansac on every keyframe, one batched launch
    d_obj = torch.as_tensor(obj, device=dev)
    d_img = torch.as_tensor(np.ascontiguousarray(feat32), device=dev)
    d_intr = torch.as_tensor(intr0, device=dev)
    d_pose = torch.empty(K, 6, dtype=torch.float64, device=dev)
    d_inl = torch.empty(K, n_circ, dtype=torch.int32, device=dev)
    d_ok = torch.empty(K, dtype=torch.int32, device=dev)
    capi.pnp_batch_dev(ctx, d_obj.data_ptr(), n_circ, d_img.data_ptr(), None, K, model, d_intr.data_ptr(), 4.0, 3, 0, d_pose.data_ptr(),
                       d_inl.data_ptr(), None, d_ok.data_ptr(), st)
    pose = d_pose.cpu().numpy()
    ok = d_ok.cpu().numpy().astype(bool)
    Rsw = rodrigues(pose[:, :3])
    tsw = pose[:, 3:]
    twb = -np.einsum("nji,nj->ni", Rsw, tsw)
    mark("pnp")
    # rectifyFeatures for all keyframes at once: their windows go through the detection pipeline again
    pipe.set_windows(kf["duration"][:, 0], kf["duration"][:, 1])
    # (every size tier at work: these windows are 4 - 10 steps long, and what the stages' previous call saw — the search's last
    # pass, its lists all but empty — would send the whole batch through the one slow general launch: 9.4 ms instead of 3)
    was_mode = ctx.get_tail_mode()      # (the caller's choice comes back afterwards, as in ecal_rectify_keyframes)
    ctx.set_tail_mode("tiered")
    try:
        pipe.run(events, eps, minpts)
    finally:
        ctx.set_tail_mode(was_mode)
    mark("rectify_detection")
    prm = capi.RectifyParams()
    prm.fx, prm.fy, prm.cx, prm.cy = intr0[:4]
    for i in range(5):
        prm.dist[i] = (intr0[5 + i] if i < 4 else 0.0) if fisheye else intr0[4 + i]   # k1 k2 p1 p2 k3 | fisheye: k1..k4 (slot 4 = alpha)
    prm.model = model
    prm.width, prm.height, prm.rows, prm.cols, prm.asymmetric = width, height, rows, cols, 1
    prm.circle_radius, prm.fit_circle = circle_radius, int(pipe.det[3])
    d_frames = torch.arange(K, dtype=torch.int32, device=dev)
    d_rpose = torch.as_tensor(np.concatenate([Rsw.reshape(K, 9), tsw], axis=1), device=dev)
    d_feat = torch.empty(K, n_circ, 3, dtype=torch.float64, device=dev)
    d_valid = torch.empty(K, n_circ, dtype=torch.int32, device=dev)
    d_info = torch.empty(K, 2, dtype=torch.int32, device=dev)
    ctx.rectify_batch_dev(pipe.xy.data_ptr(), pipe.seg_off.data_ptr(), pipe.seg_cnt.data_ptr(), pipe.kept_labels.data_ptr(),
                          pipe.win_info.data_ptr(), d_frames.data_ptr(), d_rpose.data_ptr(), K, d_obj.data_ptr(), prm,
                          d_feat.data_ptr(), d_valid.data_ptr(), d_info.data_ptr(), st)
    rect_ok = d_info[:, 0].cpu().numpy().astype(bool)
    mark("rectify")
    # the sequential gates of EventCalibIni.cpp:281-302 (checkPose against the last accepted keyframe, then rectify): the loop
    # carries a dependence from frame to frame — host code of the library (ecal_pose_gates), the same operations as
    # EventCalibIni::checkPose in the shim (host/event_calib_ini.hpp); 6 ms as a Python loop on 9227 keyframes
    kt = kf["time"]
    acc, n_check, n_rect = capi.pose_gates(Rsw, twb, kt, ok, rect_ok, step)
    out["init"].update(accepted=len(acc), discarded_by_check_pose=n_check, discarded_by_rectify=n_rect)
    if tables:
        out["kf"] = kf
        out["pose"] = dict(Rsw=Rsw, tsw=tsw, ok=ok, rect_ok=rect_ok, rect=d_feat.cpu().numpy())
        out["accepted"] = np.asarray(acc)
    mark("check_pose_gates")
    if len(acc) <= 10:
        raise RuntimeError("too few frames in the map.")     # EventCalibSpline.cpp:26-28
    # -- 3. splines: a gap of more than 50 steps starts a new one; fewer than 4 frames -> dropped (:318-345)
    times = kf["time"][acc]
    cuts = np.nonzero(np.diff(times) > 50 * step)[0] + 1
    segs = [s for s in np.split(np.arange(len(acc)), cuts) if len(s) >= 4]
    if not segs:
        raise RuntimeError("sampleSets not filtered")
    Qwb = quat_from_matrix(np.transpose(Rsw[acc], (0, 2, 1)))
    seg_cp_off, knots, cq, ct, ranges, jobs = [0], [], [], [], [], []
    for s in segs:
        u = times[s].copy()
        u[0] -= 3 * step
        u[-1] += 3 * step
        cp_num = int(np.floor((u[-1] - u[0]) / (50 * step)))
        if cp_num > len(u):
            cp_num = len(u) - 1
        cp_num = max(cp_num, 4)
        jobs.append((u, s, cp_num))
        seg_cp_off.append(seg_cp_off[-1] + cp_num)
        ranges.append((u[0], u[-1]))
    twb_acc = twb[acc]

    def fit(job):   # (host code of libecal.so: ctypes releases the interpreter lock, the segments' fits run side by side)
        u, s, cp_num = job
        kn, c_t = capi.spline_fit(u, twb_acc[s], cp_num)
        _, c_q = capi.spline_fit(u, Qwb[s], cp_num)
        if use_so3:   # BsplineSO3's constructor: unit quaternions, then optimizeCP (BsplineSO3.cpp:55, :285-341)
            c_q /= np.linalg.norm(c_q, axis=1, keepdims=True)
            c_q, _ = capi.spline_so3_refine(kn, c_q, Qwb[s], u)
        return kn, c_q, c_t
    if len(jobs) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), 8)) as pool:
            fits = list(pool.map(fit, jobs))
    else:
        fits = [fit(j) for j in jobs]
    for kn, c_q, c_t in fits:
        knots.append(kn)
        cq.append(c_q)
        ct.append(c_t)
    keep = np.concatenate(segs)
    kf_idx = acc[keep]
    mark("spline_fit")
    # association (EventCalibSpline.cpp:140-192): every event against the nearest keyframe's rectified circles — all spline
    # segments in ONE pass over the stream, the residual arrays and their count stay in HBM and the solver is built on them
    # in place (ecal_associate_ranges_dev -> ecal_solver_create_dev): nothing proportional to the events crosses PCIe
    n_events = events.numel() // 25
    d_kidx = torch.as_tensor(kf_idx, device=dev)
    d_kt = torch.as_tensor(kf["time"][kf_idx], device=dev)
    d_kc = d_feat[d_kidx].contiguous()                       # the rectified circles of the kept keyframes, never downloaded
    d_rng = torch.as_tensor(np.ascontiguousarray(np.array(ranges, np.float64)), device=dev)
    d_o = torch.empty(n_events, 2, dtype=torch.float64, device=dev)
    d_t = torch.empty(n_events, dtype=torch.float64, device=dev)
    d_l = torch.empty(n_events, dtype=torch.int32, device=dev)
    d_s = torch.empty(n_events, dtype=torch.int32, device=dev)
    d_c = torch.zeros(1, dtype=torch.int32, device=dev)
    ctx.associate_ranges_dev(events.data_ptr(), n_events, d_kt.data_ptr(), d_kc.data_ptr(), len(kf_idx), n_circ, d_rng.data_ptr(),
                             len(ranges), 5 * step, 5.0, d_o.data_ptr(), d_t.data_ptr(), d_l.data_ptr(), d_s.data_ptr(),
                             d_c.data_ptr(), st)
    mark("association")
    # intrinsics: K + the inverse radial polynomial of (k1, k2, k3) (:93-105)
    b5 = capi.inverse_radial_distortion([intr0[5], intr0[6], intr0[7], intr0[8]] if fisheye else [intr0[4], intr0[5], intr0[8], 0.0])
    x0 = np.concatenate([intr0[:4], b5, np.concatenate(cq).ravel(), np.concatenate(ct).ravel()])
    prob = dict(seg_cp_off=np.array(seg_cp_off, np.uint32), knots=np.concatenate(knots),
                landmarks=board_points(rows, cols, square).astype(np.float64), circle_radius=circle_radius,
                huber_a=0.2 * circle_radius, use_so3=bool(use_so3), fisheye=bool(fisheye))
    solver = capi.Solver(ctx, prob, device_arrays=(d_o.data_ptr(), d_t.data_ptr(), d_l.data_ptr(), d_s.data_ptr(), n_events,
                                                   d_c.data_ptr()), stream=st)
    n_res = solver.n_res
    if tables:
        m = int(d_c.item())
        out["spline_start"] = dict(knots=prob["knots"], x0=x0, residuals=m, obs=d_o[:m].cpu().numpy(), time=d_t[:m].cpu().numpy(),
                                   lm_id=d_l[:m].cpu().numpy())
    del d_o, d_t, d_l, d_s
    mark("solver_setup")
    opt = solver.default_options()
    opt.max_num_iterations = max_num_iterations
    x, summ = solver.solve(x0, opt)
    solver.close()
    mark("lm_solve")
    out["spline"] = {"splines": len(segs), "control_points": int(seg_cp_off[-1]), "residuals": int(n_res),
                     "iterations": int(summ.iterations), "initial_cost": float(summ.initial_cost),
                     "final_cost": float(summ.final_cost), "seconds": float(summ.seconds),
                     "jacobian_evaluations": int(summ.jacobian_evaluations), "unknowns": int(9 + 6 * seg_cp_off[-1])}
    out["intrinsics"] = x[:9].copy()
    # updateMap (:253-317): keyframe poses re-read from the optimised splines
    n_cp = seg_cp_off[-1]
    q_all, t_all = x[9:9 + 4 * n_cp].reshape(n_cp, 4), x[9 + 4 * n_cp:].reshape(n_cp, 3)
    traj = []
    for i, s in enumerate(segs):
        a, b = seg_cp_off[i], seg_cp_off[i + 1]
        tt = times[s]
        q = capi.spline_eval(knots[i], q_all[a:b], tt)
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        traj.append(np.concatenate([tt[:, None], capi.spline_eval(knots[i], t_all[a:b], tt), q], axis=1))
    out["trajectory"] = np.concatenate(traj)                 # timestamp tx ty tz qx qy qz qw (TUM)
    out["init_trajectory"] = np.concatenate([times[keep][:, None], twb[kf_idx], Qwb[keep]], axis=1)
    mark("update_map")
    return out


def release_scratch(ctx):
    """Drop the detection pipeline calibrate_stream keeps on the context (its HBM arrays go back to torch's allocator)."""
    pipe = getattr(ctx, "_calibrate_pipe", None)
    if pipe is not None:
        try:
            del ctx._calibrate_pipe
        except AttributeError:
            pass
        del pipe
        torch.cuda.empty_cache()


def save_trajectory_tum(path, traj):
    """SystemBase::saveKeyFrameTrajectoryTUM's format: fixed, 10 digits, one keyframe per line."""
    with open(path, "w") as f:
        for r in traj:
            f.write(" ".join("%.10f" % v for v in r) + "\n")
