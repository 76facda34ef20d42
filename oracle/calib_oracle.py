"""TEST INFRASTRUCTURE ONLY — CPU (numpy) restatement of the init calibration the reference delegates to OpenCV.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product path
(libecal.so) never does.

What it restates (reference call sites, paths relative to the reference tree):
  * cv::calibrateCamera(objectPoints, imagePoints, imageSize, K, dist(8), rvecs, tvecs, flag | CALIB_USE_LU)
    event_camera_calib/src/EventCalibIni.cpp:198-199, flags assembled in parameters.hpp:48-58
  * cv::fisheye::calibrate(..., flag)                                  EventCalibIni.cpp:188, parameters.hpp:60-68
  * cv::solvePnPRansac(obj, img, K, dist, rvec, tvec, false, 50, 4.0, 0.99, inliers, SOLVEPNP_IPPE)   :258-259
  * cv::projectPoints / cv::fisheye::projectPoints                      :127-130, CirclesEventFrame.cpp:449
  * cv::Rodrigues                                                       :266

PARITY UNPINNED: OpenCV (>= 4.0, CMakeLists.txt:43, no version pinned, not vendored) is absent from /root/reference
and from this image, and the reference holds no golden vectors for these calls.  The functions below follow
OpenCV's published algorithms — Zhang's closed form restricted to the focal lengths (cvInitIntrinsicParams2D),
homography -> pose, the CvLevMarq state machine (lambda = 10^-3 start, x10 / /10, diagonal scaled by 1 + lambda,
TermCriteria(30, DBL_EPSILON)), the fisheye smoothed Gauss-Newton (alpha_smooth 0.4, TermCriteria(100,
DBL_EPSILON), CALIB_RECOMPUTE_EXTRINSIC), IPPE (Collins & Bartoli 2014) — and are pinned by synthetic ground
truth (known K / distortion / poses), by finite-difference checks of every Jacobian and by the independent
scipy.optimize.least_squares minimum in tests/test_oracle_calib.py.
"""
import numpy as np

# flag bits of ecal_calib_options.flags (include/ecal.h); they mirror the cv::CALIB_* bits the reference sets
FIX_ASPECT_RATIO = 1 << 0
FIX_PRINCIPAL_POINT = 1 << 1
ZERO_TANGENT_DIST = 1 << 2
FIX_K1, FIX_K2, FIX_K3, FIX_K4, FIX_K5, FIX_K6 = (1 << 3), (1 << 4), (1 << 5), (1 << 6), (1 << 7), (1 << 8)
FIX_SKEW = 1 << 9                 # fisheye only
RECOMPUTE_EXTRINSIC = 1 << 10     # fisheye only

NI = 12  # intrinsics slots: pinhole fx fy cx cy k1 k2 p1 p2 k3 k4 k5 k6 ; fisheye fx fy cx cy alpha k1 k2 k3 k4


def rodrigues(v):
    """rvec -> R (cv::Rodrigues)."""
    v = np.asarray(v, float)
    th = np.linalg.norm(v)
    if th < np.finfo(float).eps:
        return np.eye(3)
    r = v / th
    c, s = np.cos(th), np.sin(th)
    K = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
    return c * np.eye(3) + (1 - c) * np.outer(r, r) + s * K


def rodrigues_inv(R):
    """R -> rvec (cv::Rodrigues on a matrix)."""
    R = np.asarray(R, float)
    U, _, Vt = np.linalg.svd(R)
    R = U @ Vt
    c = (np.trace(R) - 1) * 0.5
    c = min(1.0, max(-1.0, c))
    th = np.arccos(c)
    ax = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.linalg.norm(ax) * 0.5
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        t = (R + np.eye(3)) * 0.5
        r = np.sqrt(np.maximum(np.diag(t), 0))
        if R[0, 1] < 0:
            r[1] = -r[1]
        if R[0, 2] < 0:
            r[2] = -r[2]
        if abs(r[0]) < abs(r[1]) and abs(r[0]) < abs(r[2]) and (R[1, 2] > 0) != (r[1] * r[2] > 0):
            r[2] = -r[2]
        return r * (th / np.linalg.norm(r))
    return ax * (0.5 * th / s)


def project(model, intr, rvec, tvec, obj):
    """Pixel coordinates [n][2] of obj [n][3].  model 0: pinhole + (k1 k2 p1 p2 k3 k4 k5 k6); 1: fisheye."""
    R = rodrigues(rvec)
    X = obj @ R.T + np.asarray(tvec, float)
    x, y = X[:, 0] / X[:, 2], X[:, 1] / X[:, 2]
    if model == 0:
        fx, fy, cx, cy, k1, k2, p1, p2, k3, k4, k5, k6 = intr[:12]
        r2 = x * x + y * y
        r4, r6 = r2 * r2, r2 * r2 * r2
        g = (1 + k1 * r2 + k2 * r4 + k3 * r6) / (1 + k4 * r2 + k5 * r4 + k6 * r6)
        xd = x * g + p1 * 2 * x * y + p2 * (r2 + 2 * x * x)
        yd = y * g + p1 * (r2 + 2 * y * y) + p2 * 2 * x * y
        return np.stack([fx * xd + cx, fy * yd + cy], 1)
    fx, fy, cx, cy, alpha, k1, k2, k3, k4 = intr[:9]
    r = np.sqrt(x * x + y * y)
    th = np.arctan(r)
    th2 = th * th
    thd = th * (1 + th2 * (k1 + th2 * (k2 + th2 * (k3 + th2 * k4))))
    sc = np.where(r > 1e-8, thd / np.where(r > 1e-8, r, 1.0), 1.0)
    xp, yp = sc * x, sc * y
    return np.stack([fx * (xp + alpha * yp) + cx, fy * yp + cy], 1)


def free_mask(model, flags):
    """1 for every intrinsics slot the optimiser may move."""
    m = np.zeros(NI)
    if model == 0:
        m[:] = 1
        if flags & FIX_ASPECT_RATIO:
            m[0] = 0
        if flags & FIX_PRINCIPAL_POINT:
            m[2] = m[3] = 0
        if flags & ZERO_TANGENT_DIST:
            m[6] = m[7] = 0
        for bit, slot in ((FIX_K1, 4), (FIX_K2, 5), (FIX_K3, 8), (FIX_K4, 9), (FIX_K5, 10), (FIX_K6, 11)):
            if flags & bit:
                m[slot] = 0
    else:
        m[:9] = 1
        if flags & FIX_PRINCIPAL_POINT:
            m[2] = m[3] = 0
        if flags & FIX_SKEW:
            m[4] = 0
        for bit, slot in ((FIX_K1, 5), (FIX_K2, 6), (FIX_K3, 7), (FIX_K4, 8)):
            if flags & bit:
                m[slot] = 0
    return m


def _unpack(model, flags, aspect, p):
    intr = p[:NI].copy()
    if model == 0 and (flags & FIX_ASPECT_RATIO):
        intr[0] = intr[1] * aspect
    return intr


def residuals(model, flags, aspect, p, obj, img):
    """Stacked projected - measured, [V][n][2] flattened; p = [intr 12 | V x (rvec 3, tvec 3)]."""
    intr = _unpack(model, flags, aspect, p)
    V = img.shape[0]
    out = np.empty((V,) + img.shape[1:])
    for v in range(V):
        q = p[NI + 6 * v: NI + 6 * v + 6]
        out[v] = project(model, intr, q[:3], q[3:], obj) - img[v]
    return out.ravel()


def jacobian_fd(model, flags, aspect, p, obj, img, h=1e-6):
    """Central-difference Jacobian of residuals() — the check of the analytic kernels, never used to solve."""
    J = np.empty((residuals(model, flags, aspect, p, obj, img).size, p.size))
    for k in range(p.size):
        d = np.zeros_like(p)
        d[k] = h * max(1.0, abs(p[k]))
        J[:, k] = (residuals(model, flags, aspect, p + d, obj, img) - residuals(model, flags, aspect, p - d, obj, img)) / (2 * d[k])
    if model == 0 and (flags & FIX_ASPECT_RATIO):
        J[:, 0] = 0
    return J


def view_blocks(model, flags, aspect, p, obj, img, v):
    """Per-view normal-equation blocks in the layout of ecal_calib_view_blocks_dev:
    Hii [12][12], Hiv [12][6], Hvv [6][6], gi [12], gv [6], cost (= sum of squared residuals of the view)."""
    m = free_mask(model, flags)
    pv = np.concatenate([p[:NI], p[NI + 6 * v: NI + 6 * v + 6]])
    Jv = jacobian_fd(model, flags, aspect, pv, obj, img[v:v + 1])
    Jv[:, :NI] *= m
    r = residuals(model, flags, aspect, pv, obj, img[v:v + 1])
    H = Jv.T @ Jv
    g = Jv.T @ r
    return H[:NI, :NI], H[:NI, NI:], H[NI:, NI:], g[:NI], g[NI:], float(r @ r)


def reduced_record(model, flags, aspect, p, obj, img, views, lam):
    """Schur-reduced record of a shard of views in the layout ranks all-reduce (ecal_calibrate_views):
    S [12][12] = sum_v Hii - Hiv (Hvv + lam diag Hvv)^-1 Hvi | g [12] | diag(sum Hii) [12] | cost | points."""
    S, g, d, cost = np.zeros((NI, NI)), np.zeros(NI), np.zeros(NI), 0.0
    for v in views:
        Hii, Hiv, Hvv, gi, gv, c = view_blocks(model, flags, aspect, p, obj, img, v)
        W = Hiv @ np.linalg.inv(Hvv + lam * np.diag(np.diag(Hvv)))
        S += Hii - W @ Hiv.T
        g += gi - W @ gv
        d += np.diag(Hii)
        cost += c
    return np.concatenate([S.ravel(), g, d, [cost, float(len(views) * obj.shape[0])]])


def homography(src, dst):
    """Least-squares homography dst ~ H src (Hartley-normalised DLT, h22 = 1)."""
    def norm(pts):
        c = pts.mean(0)
        d = np.sqrt(((pts - c) ** 2).sum(1)).mean()
        s = np.sqrt(2.0) / d
        return np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1]])
    Ts, Td = norm(src), norm(dst)
    a = (np.c_[src, np.ones(len(src))] @ Ts.T)[:, :2]
    b = (np.c_[dst, np.ones(len(dst))] @ Td.T)[:, :2]
    A = np.zeros((2 * len(a), 8))
    rhs = np.zeros(2 * len(a))
    A[0::2, 0:2], A[0::2, 2], A[0::2, 6:8] = a, 1, -b[:, :1] * a
    A[1::2, 3:5], A[1::2, 5], A[1::2, 6:8] = a, 1, -b[:, 1:] * a
    rhs[0::2], rhs[1::2] = b[:, 0], b[:, 1]
    h = np.linalg.solve(A.T @ A, A.T @ rhs)
    Hn = np.append(h, 1.0).reshape(3, 3)
    H = np.linalg.inv(Td) @ Hn @ Ts
    return H / H[2, 2]


def undistort_normalized(model, intr, px):
    """Pixels -> ideal normalised coordinates (cv::undistortPoints: 5 fixed-point iterations; fisheye: Newton on
    theta, 10 iterations)."""
    if model == 0:
        fx, fy, cx, cy, k1, k2, p1, p2, k3, k4, k5, k6 = intr[:12]
        x0, y0 = (px[:, 0] - cx) / fx, (px[:, 1] - cy) / fy
        x, y = x0.copy(), y0.copy()
        for _ in range(5):
            r2 = x * x + y * y
            ic = (1 + ((k6 * r2 + k5) * r2 + k4) * r2) / (1 + ((k3 * r2 + k2) * r2 + k1) * r2)
            dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
            dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
            x, y = (x0 - dx) * ic, (y0 - dy) * ic
        return np.stack([x, y], 1)
    fx, fy, cx, cy, alpha, k1, k2, k3, k4 = intr[:9]
    yp = (px[:, 1] - cy) / fy
    xp = (px[:, 0] - cx) / fx - alpha * yp
    thd = np.sqrt(xp * xp + yp * yp)
    thd = np.clip(thd, -np.pi / 2, np.pi / 2)
    th = thd.copy()
    for _ in range(10):
        t2 = th * th
        f = th * (1 + t2 * (k1 + t2 * (k2 + t2 * (k3 + t2 * k4)))) - thd
        df = 1 + t2 * (3 * k1 + t2 * (5 * k2 + t2 * (7 * k3 + t2 * 9 * k4)))
        th = th - f / df
    sc = np.where(thd > 1e-8, np.tan(th) / np.where(thd > 1e-8, thd, 1.0), 1.0)
    return np.stack([xp * sc, yp * sc], 1)


def ippe(obj, nrm):
    """Planar pose from normalised image points (Collins & Bartoli 2014, "Infinitesimal plane-based pose
    estimation" — the SOLVEPNP_IPPE solver): both solutions, best first.  obj [n][3] with z == 0."""
    c = obj[:, :2].mean(0)
    H = homography(obj[:, :2] - c, nrm)
    p, q = H[0, 2], H[1, 2]
    J = np.array([[H[0, 0] - H[2, 0] * p, H[0, 1] - H[2, 1] * p], [H[1, 0] - H[2, 0] * q, H[1, 1] - H[2, 1] * q]])
    t = np.sqrt(p * p + q * q + 1)
    w = np.array([p, q, 1.0]) / t
    k = np.array([-w[1], w[0], 0.0])
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    Rv = np.eye(3) + Kx + Kx @ Kx / (1 + w[2])
    B = np.array([[1, 0, -p], [0, 1, -q]]) @ Rv
    A = np.linalg.solve(B[:, :2], J)
    AtA = A.T @ A
    T, D = AtA[0, 0] + AtA[1, 1], AtA[0, 0] * AtA[1, 1] - AtA[0, 1] * AtA[1, 0]
    gamma = np.sqrt(0.5 * (T + np.sqrt(max(T * T - 4 * D, 0.0))))
    R22 = A / gamma
    b1 = np.sqrt(max(0.0, 1 - R22[0, 0] ** 2 - R22[1, 0] ** 2))
    b2 = np.sqrt(max(0.0, 1 - R22[0, 1] ** 2 - R22[1, 1] ** 2))
    if R22[0, 0] * R22[0, 1] + R22[1, 0] * R22[1, 1] > 0:
        b2 = -b2
    sols = []
    X = np.c_[obj[:, :2] - c, np.zeros(len(obj))]
    for sgn in (1.0, -1.0):
        r1 = np.array([R22[0, 0], R22[1, 0], sgn * b1])
        r2 = np.array([R22[0, 1], R22[1, 1], sgn * b2])
        R = Rv @ np.stack([r1, r2, np.cross(r1, r2)], 1)
        P = X @ R.T
        A3 = np.zeros((2 * len(X), 3))
        rhs = np.zeros(2 * len(X))
        A3[0::2, 0], A3[0::2, 2], rhs[0::2] = 1, -nrm[:, 0], nrm[:, 0] * P[:, 2] - P[:, 0]
        A3[1::2, 1], A3[1::2, 2], rhs[1::2] = 1, -nrm[:, 1], nrm[:, 1] * P[:, 2] - P[:, 1]
        tv = np.linalg.solve(A3.T @ A3, A3.T @ rhs)
        Y = P + tv
        err = float((((Y[:, :2] / Y[:, 2:3]) - nrm) ** 2).sum())
        sols.append((err, R, tv - R @ np.array([c[0], c[1], 0.0])))
    sols.sort(key=lambda s: s[0])
    return sols


def levmarq(fun_blocks, fun_err, p0, mask, max_iter=30, eps=np.finfo(float).eps):
    """The CvLevMarq::updateAlt state machine (OpenCV calib3d, used by calibrateCamera and
    cvFindExtrinsicCameraParams2): fun_blocks(p) -> (JtJ, JtErr, errNorm); fun_err(p) -> errNorm."""
    p = p0.copy()
    lam_lg10 = -3
    idx = np.flatnonzero(mask)
    it = 0
    JtJ, JtErr, err = fun_blocks(p)
    while True:
        prev, prev_err = p.copy(), err
        while True:                                   # step(); CHECK_ERR
            lam = 10.0 ** lam_lg10
            A = JtJ[np.ix_(idx, idx)].copy()
            A[np.diag_indices_from(A)] *= 1 + lam
            x = np.linalg.solve(A, JtErr[idx])
            p = prev.copy()
            p[idx] -= x
            err = fun_err(p)
            if err > prev_err:
                lam_lg10 += 1
                if lam_lg10 <= 16:
                    continue
            break
        lam_lg10 = max(lam_lg10 - 1, -16)
        it += 1
        if it >= max_iter or np.linalg.norm(p - prev) / np.linalg.norm(prev) < eps:
            return p, err, it
        JtJ, JtErr, err = fun_blocks(p)


def refine_pose(model, intr, obj, img, rvec, tvec, max_iter=20, eps=np.finfo(np.float32).eps):
    """LM refinement of one pose (cvFindExtrinsicCameraParams2's CvLevMarq(6, ..., 20, FLT_EPSILON))."""
    def blocks(q):
        r = (project(model, intr, q[:3], q[3:], obj) - img).ravel()
        J = np.empty((r.size, 6))
        for k in range(6):
            d = np.zeros(6)
            d[k] = 1e-6 * max(1.0, abs(q[k]))
            J[:, k] = ((project(model, intr, (q + d)[:3], (q + d)[3:], obj) - project(model, intr, (q - d)[:3], (q - d)[3:], obj)).ravel()) / (2 * d[k])
        return J.T @ J, J.T @ r, float(r @ r)

    def err(q):
        r = (project(model, intr, q[:3], q[3:], obj) - img).ravel()
        return float(r @ r)
    q, e, _ = levmarq(blocks, err, np.concatenate([rvec, tvec]), np.ones(6), max_iter, eps)
    return q[:3], q[3:], e


def view_pose(model, intr, obj, img, refine_iters=20):
    """Initial pose of one view: undistort -> IPPE -> (optional) LM refinement."""
    nrm = undistort_normalized(model, intr, img)
    _, R, t = ippe(obj, nrm)[0]
    rvec = rodrigues_inv(R)
    if refine_iters > 0:
        rvec, t, _ = refine_pose(model, intr, obj, img, rvec, t, refine_iters)
    return rvec, t


def init_focal(obj, img, width, height, aspect):
    """cvInitIntrinsicParams2D: principal point at the image centre, focal lengths from the vanishing-point
    constraints of every view's homography (Zhang 2000, restricted to fx, fy)."""
    cx, cy = (width - 1) * 0.5, (height - 1) * 0.5
    A, b = [], []
    for v in range(img.shape[0]):
        H = homography(obj[:, :2], img[v])
        H[0] -= H[2] * cx
        H[1] -= H[2] * cy
        h, vv = H[:, 0].copy(), H[:, 1].copy()
        d1, d2 = (h + vv) * 0.5, (h - vv) * 0.5
        n = [np.linalg.norm(z) for z in (h, vv, d1, d2)]
        h, vv, d1, d2 = h / n[0], vv / n[1], d1 / n[2], d2 / n[3]
        A += [[h[0] * vv[0], h[1] * vv[1]], [d1[0] * d2[0], d1[1] * d2[1]]]
        b += [-h[2] * vv[2], -d1[2] * d2[2]]
    A, b = np.array(A), np.array(b)
    f = np.linalg.solve(A.T @ A, A.T @ b)
    fx, fy = np.sqrt(abs(1 / f[0])), np.sqrt(abs(1 / f[1]))
    if aspect != 0:
        tf = (fx + fy) / (aspect + 1)
        fx, fy = aspect * tf, tf
    return fx, fy, cx, cy


def calibrate(model, obj, img, width, height, flags, aspect=0.0, max_iter=None, eps=np.finfo(float).eps):
    """calibrateCamera (model 0) / fisheye::calibrate (model 1).  img [V][n][2], obj [n][3] (z == 0).
    Returns intr [12], rvecs [V][3], tvecs [V][3], rms, iterations."""
    V = img.shape[0]
    mask = free_mask(model, flags)
    intr = np.zeros(NI)
    if model == 0:
        a = aspect if (flags & FIX_ASPECT_RATIO) else 0.0
        intr[:4] = init_focal(obj, img, width, height, a)
    else:
        f = max(width, height) / np.pi
        intr[:4] = f, f, width / 2.0 - 0.5, height / 2.0 - 0.5
    poses = [view_pose(model, intr, obj, img[v]) for v in range(V)]
    p = np.concatenate([intr] + [np.concatenate(q) for q in poses])
    full_mask = np.concatenate([mask, np.ones(6 * V)])

    def blocks(q):
        J = jacobian_fd(model, flags, aspect, q, obj, img) * full_mask
        r = residuals(model, flags, aspect, q, obj, img)
        return J.T @ J, J.T @ r, float(r @ r)

    def err(q):
        r = residuals(model, flags, aspect, q, obj, img)
        return float(r @ r)

    if model == 0:
        p, e, it = levmarq(blocks, err, p, full_mask, 30 if max_iter is None else max_iter, eps)
        if flags & FIX_ASPECT_RATIO:
            p[0] = p[1] * aspect
    else:
        idx = np.flatnonzero(full_mask)
        change, it = 1.0, 0
        mi = 100 if max_iter is None else max_iter
        while it < mi and change > eps:
            JtJ, JtErr, _ = blocks(p)
            a2 = 1 - (1 - 0.4) ** (it + 1)
            x = np.linalg.solve(JtJ[np.ix_(idx, idx)], JtErr[idx])
            q = p.copy()
            q[idx] -= a2 * x
            change = np.linalg.norm(q[:4] - p[:4]) / np.linalg.norm(q[:4])
            p = q
            if flags & RECOMPUTE_EXTRINSIC:
                for v in range(V):
                    rv, tv = view_pose(model, p[:NI], obj, img[v])
                    p[NI + 6 * v: NI + 6 * v + 6] = np.concatenate([rv, tv])
            it += 1
        e = err(p)
    rms = np.sqrt(e / (V * obj.shape[0]))
    return p[:NI], p[NI:].reshape(V, 6)[:, :3], p[NI:].reshape(V, 6)[:, 3:], rms, it


def pnp_consensus(model, intr, obj, img, thresh=4.0, rounds=3):
    """Deterministic stand-in for solvePnPRansac(..., 50, 4.0, 0.99, inliers, SOLVEPNP_IPPE): IPPE on the current
    inlier set, inliers = reprojection error <= thresh px, repeated until the set is stable (<= rounds)."""
    inl = np.ones(len(obj), bool)
    rvec = t = None
    for _ in range(rounds):
        if inl.sum() < 4:
            return None, None, inl
        nrm = undistort_normalized(model, intr, img[inl])
        _, R, t = ippe(obj[inl], nrm)[0]
        rvec = rodrigues_inv(R)
        e = np.sqrt(((project(model, intr, rvec, t, obj) - img) ** 2).sum(1))
        new = e <= thresh
        if (new == inl).all():
            break
        inl = new
    return rvec, t, inl
