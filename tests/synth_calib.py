"""Synthetic calibration views for the init-calibration path (SURVEY §8d ground-truth camera, §8e row 3)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import calib_oracle as CO  # noqa: E402

WIDTH, HEIGHT = 346.0, 260.0
GT_PINHOLE = np.array([359.67525, 359.67525, 172.5, 129.5, -0.34991902, -0.014698517, 0.0, 0.0, 0.59684463, 0, 0, 0])
# a real fisheye: f ~ width / pi (the start value of cv::fisheye::calibrate), ~160 deg across the image
GT_FISHEYE = np.array([125.0, 125.0, 172.5, 129.5, 0.0, 0.05, -0.01, 0.002, 0.0, 0, 0, 0])
# the shipped example.yaml: fixed aspect ratio 1, principal point at the centre, no tangential, K4..K6 fixed
FLAGS_EXAMPLE = (CO.FIX_ASPECT_RATIO | CO.FIX_PRINCIPAL_POINT | CO.ZERO_TANGENT_DIST | CO.FIX_K4 | CO.FIX_K5 | CO.FIX_K6)
FLAGS_FISHEYE = CO.FIX_SKEW | CO.RECOMPUTE_EXTRINSIC | CO.FIX_K4


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
        R = CO.rodrigues(rvec)
        if model == 0:
            centre = np.array([rng.uniform(-12, 12), rng.uniform(-8, 8), rng.uniform(50, 80)])
        else:
            centre = np.array([rng.uniform(-8, 8), rng.uniform(-6, 6), rng.uniform(18, 30)])
        tvec = centre - R @ c
        px = CO.project(model, intr, rvec, tvec, obj)
        if px[:, 0].min() < 4 or px[:, 0].max() > WIDTH - 5 or px[:, 1].min() < 4 or px[:, 1].max() > HEIGHT - 5:
            continue
        img.append(px + noise_px * rng.normal(size=px.shape))
        rv.append(rvec)
        tv.append(tvec)
    return obj, np.array(img), np.array(rv), np.array(tv)
