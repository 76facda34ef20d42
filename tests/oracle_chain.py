"""TEST INFRASTRUCTURE ONLY — the reference driver end to end ON THE CPU, assembled from the stage oracles.

What unit_test_eventCameraCalib computes on a .bin stream (event_camera_calib/test/eventCameraCalib.cpp:99-233), stage by
stage, each stage by the oracle that restates it — nothing of libecal.so is called here:

  1. keyframe search    oracle/policy_oracle.cpp (MultiProcess::process + EventCalibIni::track, eventCameraCalib.cpp:34-97,
                        EventCalibIni.cpp:23-97), every window through oracle/event_oracle.cpp (EventFrame, reference element
                        order), oracle/dbscan_oracle.cpp on the reference's kd-tree when oracle/_ref is built, and
                        oracle/detect_oracle.cpp (extractFeatures with the reference's nth_element picks).  The ordering of
                        the candidates into the pattern is the ONE stage without a CPU restatement: the reference's finder is
                        OpenCV's randomised findCirclesGrid (CirclesEventFrame.cpp:332-336) followed by a nearest-candidate
                        lookup without a distance bound (:340-353) — absent here and unpinnable.  `grid_order` plugs it in:
                        gt_grid_order (ground truth of a synthetic stream: a window holds the grid iff every circle has its
                        own candidate within GT_TOL_PX) for CPU-only runs, or the product's ecal_grid_order_dev on the
                        oracle's candidates (tests/test_gpu_oracle_chain.py, which also states how far the two agree)
  2. init calibration   oracle/calib_oracle.py calibrate (calibrateCamera's loop, EventCalibIni.cpp:163-199), pnp_consensus for
                        every keyframe (:258-259), checkPose (:327-347), oracle/rectify_oracle.cpp (rectifyFeatures,
                        CirclesEventFrame.cpp:417-638), the sequential gates of :281-302
  3. splines            gap segmentation and control-point counts (EventCalibSpline.cpp:61-91, :319-345),
                        oracle/spline_fit_oracle.py (BsplineReal.hpp:329-449), oracle/associate_oracle.cpp (:140-192),
                        inverseRadialDistortion (PinholeCamera.cpp:74-92), Levenberg-Marquardt on oracle/solver_oracle.cpp's
                        dual-number normal equations (the loop of tests/ref_lm.py on the threaded arrow-layout evaluation)
  4. updateMap          keyframe poses re-read from the optimised splines (:253-317)

tests/test_gpu_oracle_chain.py compares eventcalib_amd.calibrate.calibrate_stream with this, stage by stage."""
import math
import os
import sys

import numpy as np

import oracle_lib as O
import ref_lm

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import calib_oracle as CO  # noqa: E402
import spline_fit_oracle as SF  # noqa: E402

RADIUS_THR = 15.511363636363637   # circleRadiusThreshold_ of example.yaml on a 346 x 260 sensor (tests/test_gpu_detect.py KAT)
GT_TOL_PX = 14.0                  # a circle's candidate lies within this of the ground-truth centre (tests/test_gpu_grid.py)


def board_points(rows=9, cols=4, square=5.5):
    """calcBoardCornerPositions (EventCalibIni.cpp:99-115), narrowed to float like cv::Point3f."""
    pts = [((2 * j + i % 2) * square, i * square, 0.0) for i in range(rows) for j in range(cols)]
    return np.array(pts, np.float32).astype(np.float64)


def grid_by_ground_truth(cand_xy, gt_xy, tol=GT_TOL_PX):
    """cand_xy [n,2], gt_xy [M,2] (model order) -> candidate index per model point, or None: a circle without a candidate
    within tol, or one candidate claimed by two circles."""
    if len(cand_xy) < len(gt_xy):
        return None
    d = np.linalg.norm(cand_xy[None, :, :] - gt_xy[:, None, :], axis=2)   # [M, n]
    pick = d.argmin(axis=1)
    if (d[np.arange(len(gt_xy)), pick] >= tol).any() or len(set(pick.tolist())) != len(gt_xy):
        return None
    return pick


def gt_grid_order(centres_at, tol=GT_TOL_PX):
    """grid_order callable from ground truth: centres_at(t) -> [M, 2] circle centres in pixels."""
    def order(cand_xyr, t_mid):
        return grid_by_ground_truth(cand_xyr[:, :2], centres_at(t_mid), tol)
    return order


class WindowOracle:
    """extractFeatures of one window on the CPU: (found, EventFrame::eventsNum(), ordered circles) + what rectifyFeatures
    needs of the same window later.  grid_order(cand_xyr [n, 3], window mid time) -> candidate index per model point or None."""

    def __init__(self, rec, grid_order, eps=4.0, minpts=2, cluster_min=5, rows=9, cols=4, radius_thr=RADIUS_THR):
        self.rec = np.ascontiguousarray(rec, np.uint8)
        self.grid_order = grid_order
        self.eps, self.minpts, self.cluster_min, self.M, self.radius_thr = eps, minpts, cluster_min, rows * cols, radius_thr
        self.cache = {}
        self.calls = 0

    def window(self, t0, t1):
        key = (float(t0), float(t1))
        if key in self.cache:
            return self.cache[key]
        self.calls += 1
        lo, hi = O.window_bounds(self.rec, t0, t1)
        n = int(hi - lo)
        out = O.detect_windows_full(self.rec, [t0], [t1], [0, n], max(n, 1), self.eps, self.minpts, self.cluster_min, self.M,
                                    self.radius_thr, n_threads=1)
        info = out["win_info"][0]
        npos, nneg = int(out["seg_cnt"][0]), int(out["seg_cnt"][1])
        res = dict(events_num=npos + nneg, found=False, features=None, pos=out["xy"][:npos].copy(),
                   neg=out["xy"][npos:npos + nneg].copy(), kept_pos=out["kept_labels"][:npos].copy(),
                   kept_neg=out["kept_labels"][npos:npos + nneg].copy(), status=int(info[3]) & 0xFF, n_cand=int(info[0]),
                   cand=out["cand_xyr"][:int(info[0])].copy(), t_mid=0.5 * (t0 + t1))
        if res["status"] == 0 and info[0] >= self.M:
            cand = res["cand"]
            pick = self.grid_order(cand, res["t_mid"])
            if pick is not None:
                res["found"] = True
                res["features"] = cand[pick].copy()
        self.cache[key] = res
        return res

    def detect(self, t0, t1):
        r = self.window(t0, t1)
        return r["found"], r["events_num"], r["features"]


def check_pose(R_ref, twb_ref, t_ref, R_cur, twb_cur, t_cur, step):
    """EventCalibIni::checkPose (EventCalibIni.cpp:327-347): translational and angular velocity between two keyframes."""
    dt = t_cur - t_ref
    v_t = np.linalg.norm(twb_cur - twb_ref) / dt
    c = (np.trace(R_cur @ R_ref.T) - 1) * 0.5
    v_r = abs(math.acos(min(1.0, max(-1.0, c))) / dt)
    return v_t < (2.5e-1 / step) * 2 and v_r < (5e-4 * math.pi) * 2 / step


def quat_from_matrix(m):
    """Eigen::Quaterniond(Matrix3d) (x y z w)."""
    tr = m[0, 0] + m[1, 1] + m[2, 2]
    if tr > 0:
        t = math.sqrt(tr + 1.0)
        w = 0.5 * t
        t = 0.5 / t
        return np.array([(m[2, 1] - m[1, 2]) * t, (m[0, 2] - m[2, 0]) * t, (m[1, 0] - m[0, 1]) * t, w])
    i = 0
    if m[1, 1] > m[0, 0]:
        i = 1
    if m[2, 2] > m[i, i]:
        i = 2
    j, k = (i + 1) % 3, (i + 2) % 3
    t = math.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0)
    q = np.zeros(4)
    q[i] = 0.5 * t
    t = 0.5 / t
    q[3] = (m[k, j] - m[j, k]) * t
    q[j] = (m[j, i] + m[i, j]) * t
    q[k] = (m[k, i] + m[i, k]) * t
    return q


def dense_from_arrow(acc, n_cp):
    """The accumulation-buffer layout (oracle_evaluate_arrow_mt / arrow_layout.hpp) -> cost, g, H in tangent order
    [intr 9 | per control point rot 3, trans 3]."""
    n = 9 + 6 * n_cp
    g = np.zeros(n)
    H = np.zeros((n, n))
    cost = acc[0]
    g[:9] = acc[1:10]
    Hi = acc[10:91].reshape(9, 9)
    H[:9, :9] = np.triu(Hi) + np.triu(Hi, 1).T
    for c in range(n_cp):
        r = acc[91 + 204 * c: 91 + 204 * (c + 1)]
        a = 9 + 6 * c
        g[a:a + 6] = r[:6]
        B = r[6:60].reshape(6, 9)                     # H_c,intr: [component][intrinsic]
        H[a:a + 6, :9] = B
        H[:9, a:a + 6] = B.T
        for d in range(4):
            if c + d >= n_cp:
                break
            blk = r[60 + 36 * d: 96 + 36 * d].reshape(6, 6)
            b = 9 + 6 * (c + d)
            if d == 0:
                blk = np.triu(blk) + np.triu(blk, 1).T
                H[a:a + 6, a:a + 6] = blk
            else:
                H[a:a + 6, b:b + 6] = blk
                H[b:b + 6, a:a + 6] = blk.T
    return cost, g, H


def evaluate_segments(problems, x, n_threads):
    """Cost, gradient and dense normal matrix of a problem of several spline segments that share the nine intrinsics:
    oracle_evaluate_arrow_mt segment by segment, scattered into tangent order [intr 9 | control points (rot 3, trans 3) of all
    segments one after the other].  x = [intr 9 | quaternions of all control points | translations of all control points]."""
    n_cps = [int(p["seg_cp_off"][-1]) for p in problems]
    n_cp = sum(n_cps)
    n = 9 + 6 * n_cp
    cost, g, H = 0.0, np.zeros(n), np.zeros((n, n))
    c0 = 0
    for p, k in zip(problems, n_cps):
        xs = np.concatenate([x[:9], x[9 + 4 * c0: 9 + 4 * (c0 + k)], x[9 + 4 * n_cp + 3 * c0: 9 + 4 * n_cp + 3 * (c0 + k)]])
        c, gs, Hs = dense_from_arrow(O.solver_evaluate_arrow(p, xs, n_threads), k)
        idx = np.concatenate([np.arange(9), 9 + 6 * c0 + np.arange(6 * k)])
        cost += c
        g[idx] += gs
        H[np.ix_(idx, idx)] += Hs
        c0 += k
    return cost, g, H


def lm_solve(problems, x0, n_threads, max_iter=50, ftol=1e-10, gtol=1e-10, ptol=1e-8):
    """tests/ref_lm.py's loop (Ceres 1.x trust-region LM, Jacobi scaling) on the threaded arrow-layout evaluation of
    oracle/solver_oracle.cpp (same sums as the dense evaluation, any problem size, any number of segments)."""
    n_cp = sum(int(p["seg_cp_off"][-1]) for p in problems)
    use_so3 = bool(problems[0].get("use_so3", False))

    def evaluate(x):
        return evaluate_segments(problems, x, n_threads)
    x = x0.copy()
    cost, g, H = evaluate(x)
    scale = 1.0 / (1.0 + np.sqrt(np.diag(H)))
    radius, dec = 1e4, 2.0
    hist = [cost]
    it = 0
    while it < max_iter:
        it += 1
        Hs = H * scale[:, None] * scale[None, :]
        dd = np.clip(np.diag(Hs), 1e-6, 1e32) / radius
        try:
            Lc = np.linalg.cholesky(Hs + np.diag(dd))
        except np.linalg.LinAlgError:
            radius /= dec
            dec *= 2
            continue
        ys = -np.linalg.solve(Lc.T, np.linalg.solve(Lc, g * scale))
        d = ys * scale
        model = -(g @ d) - 0.5 * d @ H @ d
        if model <= 0:
            radius /= dec
            dec *= 2
            continue
        xn = ref_lm.plus(x, d, n_cp, use_so3)
        new_cost, gn, Hn = evaluate(xn)
        rel = (cost - new_cost) / model
        if rel > 1e-3:
            change, prev = cost - new_cost, cost
            x, cost, g, H = xn, new_cost, gn, Hn
            hist.append(cost)
            t = 2 * rel - 1
            radius = min(1e16, radius / max(1.0 / 3.0, 1 - t ** 3))
            dec = 2.0
            if np.abs(g).max() <= gtol or abs(change) <= ftol * prev:
                break
        else:
            radius /= dec
            dec *= 2
        if np.linalg.norm(d) <= ptol * (np.linalg.norm(x) + ptol):
            break
    return x, hist, it


def run_chain(rec, grid_order, t_first, t_last, *, motion_time_step=5e-4, frame_event_num_threshold=4000, piece_num=30,
              frames_to_use=200, width=346.0, height=260.0, rows=9, cols=4, square=5.5, circle_radius=1.75, flags=None,
              aspect_ratio=1.0, use_so3=False, max_num_iterations=50, eps=4.0, minpts=2, n_threads=None, log=None):
    """rec: the packed 25-byte records (numpy uint8); grid_order: see WindowOracle (gt_grid_order(centres_at) on the CPU).
    Returns a dict shaped like eventcalib_amd.calibrate.calibrate_stream's (keyframes, init, spline, intrinsics, trajectory)
    plus the per-stage tables the parity test compares."""
    if flags is None:
        flags = CO.FIX_ASPECT_RATIO | CO.FIX_PRINCIPAL_POINT | CO.ZERO_TANGENT_DIST | CO.FIX_K4 | CO.FIX_K5 | CO.FIX_K6
    if n_threads is None:
        n_threads = max(1, (os.cpu_count() or 1))
    say = log or (lambda *a: None)
    step, M = motion_time_step, rows * cols
    W = WindowOracle(rec, grid_order, eps, minpts, 5, rows, cols)
    # -- 1. keyframes: the reference's single-worker run (one keyframe map for all pieces)
    kf = O.policy_run(W.detect, t_first, t_last, piece_num, step, frame_event_num_threshold, rows, cols, mode=1)
    K = len(kf["time"])
    out = dict(keyframes=K, kf=kf, windows_evaluated=W.calls, window_oracle=W)
    say("keyframes", K, "windows", kf["windows"], "evaluated", W.calls)
    if K == 0:
        raise RuntimeError("no keyframe found")
    # -- 2. init calibration on a subset (EventCalibIni.cpp:163-181); image points narrowed to float like cv::Point2f
    use = frames_to_use
    sel_step = K // use
    if sel_step == 0:
        use, sel_step = K, 1
    sel = np.arange(use) * sel_step
    obj = board_points(rows, cols, square)
    feat32 = kf["features"][:, :, :2].astype(np.float32).astype(np.float64)
    intr0, rvs, tvs, rms, iters = CO.calibrate(0, obj, feat32[sel], width, height, flags, aspect_ratio)
    out["init"] = dict(intr=intr0, rms=rms, iterations=iters, views=use)
    say("init", intr0[:9], "rms", rms, "iterations", iters)
    Rsw = np.zeros((K, 3, 3))
    tsw = np.zeros((K, 3))
    ok = np.zeros(K, bool)
    for f in range(K):
        rv, tv, _ = CO.pnp_consensus(0, intr0, obj, feat32[f], 4.0, 3)
        if rv is not None:
            ok[f] = True
            Rsw[f] = CO.rodrigues(rv)
            tsw[f] = tv
    twb = -np.einsum("nji,nj->ni", Rsw, tsw)
    # rectifyFeatures for every keyframe (its window through the detection oracle again)
    dist = np.array([intr0[4], intr0[5], intr0[6], intr0[7], intr0[8]])      # k1 k2 p1 p2 k3
    rect = np.full((K, M, 3), np.nan)
    rect_ok = np.zeros(K, bool)
    for f in range(K):
        w = W.window(kf["duration"][f, 0], kf["duration"][f, 1])
        pose = np.concatenate([Rsw[f].ravel(), tsw[f]])
        feat, valid, okf, erased = O.rectify(w["pos"], w["neg"], w["kept_pos"], w["kept_neg"], pose, intr0[:4], dist, width, height,
                                             obj, rows, cols, True, circle_radius, fit_circle=False, model=0)
        rect[f] = feat
        rect_ok[f] = bool(okf)
    out["pose"] = dict(Rsw=Rsw, tsw=tsw, ok=ok, rect_ok=rect_ok, rect=rect)
    # the sequential gates of EventCalibIni.cpp:281-302: PnP success, checkPose against the last accepted keyframe, rectify
    acc, n_check, n_rect = [], 0, 0
    last = -1
    for f in range(K):
        if not ok[f]:
            continue
        if last >= 0 and not check_pose(Rsw[last].T, twb[last], kf["time"][last], Rsw[f].T, twb[f], kf["time"][f], step):
            n_check += 1
            continue
        if not rect_ok[f]:
            n_rect += 1
            continue
        acc.append(f)
        last = f
    acc = np.array(acc, np.int64)
    out["init"].update(accepted=len(acc), discarded_by_check_pose=n_check, discarded_by_rectify=n_rect)
    out["accepted"] = acc
    say("accepted", len(acc), "checkPose", n_check, "rectify", n_rect)
    if len(acc) <= 10:
        raise RuntimeError("too few frames in the map.")
    # -- 3. splines
    times = kf["time"][acc]
    cuts = np.nonzero(np.diff(times) > 50 * step)[0] + 1
    segs = [s for s in np.split(np.arange(len(acc)), cuts) if len(s) >= 4]
    if not segs:
        raise RuntimeError("sampleSets not filtered")
    Qwb = np.array([quat_from_matrix(Rsw[f].T) for f in acc])
    twb_acc = twb[acc]
    lms = board_points(rows, cols, square)
    problems, knots, cqs, cts, cp_nums = [], [], [], [], []
    obs_all, tm_all, lm_all, seg_all = [], [], [], []
    for si, s in enumerate(segs):
        u = times[s].copy()
        u[0] -= 3 * step
        u[-1] += 3 * step
        cp_num = int(np.floor((u[-1] - u[0]) / (50 * step)))
        if cp_num > len(u):
            cp_num = len(u) - 1
        cp_num = max(cp_num, 4)
        kn, c_t = SF.fit(u, twb_acc[s], cp_num)
        _, c_q = SF.fit(u, Qwb[s], cp_num)
        kf_idx = acc[s]
        obs, tm, lm = O.associate(rec, kf["time"][kf_idx], rect[kf_idx], u[0], u[-1], 5 * step, 5.0)
        problems.append(dict(seg_cp_off=np.array([0, cp_num], np.uint32), knots=kn, obs=obs, time=tm, lm_id=lm, landmarks=lms,
                             circle_radius=circle_radius, huber_a=0.2 * circle_radius, use_so3=bool(use_so3)))
        knots.append(kn)
        cqs.append(c_q)
        cts.append(c_t)
        cp_nums.append(cp_num)
        obs_all.append(obs)
        tm_all.append(tm)
        lm_all.append(lm)
        seg_all.append(np.full(len(tm), si, np.int64))
    n_cp = sum(cp_nums)
    b5 = O.inverse_radial([intr0[4], intr0[5], intr0[8], 0.0])
    x0 = np.concatenate([intr0[:4], b5, np.concatenate(cqs).ravel(), np.concatenate(cts).ravel()])
    n_res = sum(len(t) for t in tm_all)
    out["spline_start"] = dict(knots=np.concatenate(knots), x0=x0, residuals=n_res, obs=np.concatenate(obs_all), time=np.concatenate(tm_all),
                               lm_id=np.concatenate(lm_all), seg=np.concatenate(seg_all), cp_nums=cp_nums)
    say("segments", len(segs), "control points", cp_nums, "residuals", n_res)
    x, hist, it = lm_solve(problems, x0, n_threads, max_iter=max_num_iterations)
    out["spline"] = dict(splines=len(segs), control_points=n_cp, residuals=n_res, iterations=it, initial_cost=hist[0], final_cost=hist[-1])
    out["intrinsics"] = x[:9].copy()
    say("refined", x[:9], "iterations", it, "cost", hist[0], "->", hist[-1])
    # -- 4. updateMap
    q_all, t_all = x[9:9 + 4 * n_cp].reshape(n_cp, 4), x[9 + 4 * n_cp:].reshape(n_cp, 3)
    traj, c0 = [], 0
    for s, kn, k in zip(segs, knots, cp_nums):
        tt = times[s]
        N = SF.design(kn, k, tt)
        q = N @ q_all[c0:c0 + k]
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        traj.append(np.concatenate([tt[:, None], N @ t_all[c0:c0 + k], q], axis=1))
        c0 += k
    out["trajectory"] = np.concatenate(traj)
    out["x"] = x
    return out
