"""Synthetic calibration views for the init-calibration path (SURVEY §8d ground-truth camera, §8e row 3)."""
import numpy as np

# ECAL_CALIB_* bits (include/ecal.h)
FIX_ASPECT_RATIO, FIX_PRINCIPAL_POINT, ZERO_TANGENT_DIST = 1 << 0, 1 << 1, 1 << 2
FIX_K1, FIX_K2, FIX_K3, FIX_K4, FIX_K5, FIX_K6, FIX_SKEW, RECOMPUTE_EXTRINSIC = (1 << 3, 1 << 4, 1 << 5, 1 << 6, 1 << 7,
                                                                                 1 << 8, 1 << 9, 1 << 10)

WIDTH, HEIGHT = 346.0, 260.0
GT_PINHOLE = np.array([359.67525, 359.67525, 172.5, 129.5, -0.34991902, -0.014698517, 0.0, 0.0, 0.59684463, 0, 0, 0])
# a real fisheye: f ~ width / pi (the start value of cv::fisheye::calibrate), ~160 deg across the image
GT_FISHEYE = np.array([125.0, 125.0, 172.5, 129.5, 0.0, 0.05, -0.01, 0.002, 0.0, 0, 0, 0])
# the shipped example.yaml: fixed aspect ratio 1, principal point at the centre, no tangential, K4..K6 fixed
FLAGS_EXAMPLE = FIX_ASPECT_RATIO | FIX_PRINCIPAL_POINT | ZERO_TANGENT_DIST | FIX_K4 | FIX_K5 | FIX_K6
FLAGS_FISHEYE = FIX_SKEW | RECOMPUTE_EXTRINSIC | FIX_K4


def rodrigues(v):
    v = np.asarray(v, float)
    th = np.linalg.norm(v)
    if th < 1e-300:
        return np.eye(3)
    r = v / th
    K = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
    return np.cos(th) * np.eye(3) + (1 - np.cos(th)) * np.outer(r, r) + np.sin(th) * K


def project(model, intr, rvec, tvec, obj):
    """The generating camera (data generator only; the checked restatement lives in oracle/calib_oracle.py)."""
    X = obj @ rodrigues(rvec).T + np.asarray(tvec, float)
    x, y = X[:, 0] / X[:, 2], X[:, 1] / X[:, 2]
    r2 = x * x + y * y
    if model == 0:
        fx, fy, cx, cy, k1, k2, p1, p2, k3, k4, k5, k6 = intr[:12]
        g = (1 + r2 * (k1 + r2 * (k2 + r2 * k3))) / (1 + r2 * (k4 + r2 * (k5 + r2 * k6)))
        xd = x * g + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        yd = y * g + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        return np.stack([fx * xd + cx, fy * yd + cy], 1)
    fx, fy, cx, cy, alpha, k1, k2, k3, k4 = intr[:9]
    r = np.sqrt(r2)
    th = np.arctan(r)
    t2 = th * th
    sc = np.where(r > 1e-8, th * (1 + t2 * (k1 + t2 * (k2 + t2 * (k3 + t2 * k4)))) / np.where(r > 1e-8, r, 1.0), 1.0)
    return np.stack([fx * (sc * x + alpha * sc * y) + cx, fy * sc * y + cy], 1)


def board(rows=9, cols=4, square=5.5, asymmetric=True):
    """calcBoardCornerPositions (EventCalibIni.cpp:99-115)."""
    pts = []
    for i in range(rows):
        for j in range(cols):
            pts.append(((2 * j + i % 2) * square, i * square, 0.0) if asymmetric else (j * square, i * square, 0.0))
    return np.array(pts)


def make_views(V, model=0, seed=0, noise_px=0.0, intr=None, obj=None):
    """V random board poses in front of the camera, all circles inside the image."""
    rng = np.random.default_rng(seed)
    obj = board() if obj is None else obj
    intr = (GT_PINHOLE if model == 0 else GT_FISHEYE) if intr is None else intr
    c = obj.mean(0)
    img, rv, tv = [], [], []
    while len(img) < V:
        ax = rng.normal(size=3)
        ax[2] *= 0.5
        ang = rng.uniform(0.05, 0.6)
        rvec = ax / np.linalg.norm(ax) * ang
        R = rodrigues(rvec)
        if model == 0:
            centre = np.array([rng.uniform(-12, 12), rng.uniform(-8, 8), rng.uniform(50, 80)])
        else:
            centre = np.array([rng.uniform(-8, 8), rng.uniform(-6, 6), rng.uniform(18, 30)])
        tvec = centre - R @ c
        px = project(model, intr, rvec, tvec, obj)
        if px[:, 0].min() < 4 or px[:, 0].max() > WIDTH - 5 or px[:, 1].min() < 4 or px[:, 1].max() > HEIGHT - 5:
            continue
        img.append(px + noise_px * rng.normal(size=px.shape))
        rv.append(rvec)
        tv.append(tvec)
    return obj, np.array(img), np.array(rv), np.array(tv)
