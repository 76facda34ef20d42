"""CPU: the numpy restatement of the init calibration (oracle/calib_oracle.py) pinned by synthetic ground truth and
by an independent minimiser (scipy.optimize.least_squares on the same residuals) — OpenCV itself is not available
(parity with it is unpinned, see the oracle's header)."""
import numpy as np
import pytest
from scipy.optimize import least_squares

import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import synth_calib as SC  # noqa: E402
import calib_oracle as CO  # noqa: E402


def test_rodrigues_roundtrip_and_known_values():
    assert np.allclose(CO.rodrigues([0, 0, 0]), np.eye(3))
    R = CO.rodrigues([0, 0, np.pi / 2])
    assert np.allclose(R, [[0, -1, 0], [1, 0, 0], [0, 0, 1]], atol=1e-15)
    rng = np.random.default_rng(0)
    for _ in range(50):
        v = rng.normal(size=3)
        v *= rng.uniform(0, 3.1) / np.linalg.norm(v)
        assert np.allclose(CO.rodrigues_inv(CO.rodrigues(v)), v, atol=1e-9)
    v = np.array([np.pi, 0, 0])                       # the theta = pi branch
    assert np.allclose(np.abs(CO.rodrigues_inv(CO.rodrigues(v))), np.abs(v), atol=1e-7)


def test_undistort_inverts_projection():
    rng = np.random.default_rng(1)
    for model, gt in ((0, SC.GT_PINHOLE), (1, SC.GT_FISHEYE)):
        xy = rng.uniform(-0.3, 0.3, size=(50, 2))
        X = np.c_[xy, np.ones(50)]
        px = CO.project(model, gt, np.zeros(3), np.zeros(3), X)
        back = CO.undistort_normalized(model, gt, px)
        assert np.abs(back - xy).max() < (2e-4 if model == 0 else 1e-10)   # pinhole: 5 fixed-point iterations only


def test_homography_and_ippe_exact_on_clean_data():
    obj, img, rv, tv = SC.make_views(6, 0, seed=2)
    for v in range(6):
        nrm = (img[v] - SC.GT_PINHOLE[2:4]) / SC.GT_PINHOLE[:2]          # ideal camera without distortion
        intr0 = np.zeros(12)
        intr0[:4] = SC.GT_PINHOLE[:4]
        px = CO.project(0, intr0, rv[v], tv[v], obj)
        nrm = (px - intr0[2:4]) / intr0[:2]
        sols = CO.ippe(obj, nrm)
        assert sols[0][0] < 1e-20 and sols[0][0] <= sols[1][0]
        assert np.allclose(CO.rodrigues_inv(sols[0][1]), rv[v], atol=1e-9)
        assert np.allclose(sols[0][2], tv[v], atol=1e-7)


def test_init_focal_close_to_truth():
    obj, img, _, _ = SC.make_views(20, 0, seed=4)
    fx, fy, cx, cy = CO.init_focal(obj, img, SC.WIDTH, SC.HEIGHT, 1.0)
    assert fx == fy and abs(fx / SC.GT_PINHOLE[0] - 1) < 0.15           # distortion is ignored by the closed form
    assert (cx, cy) == ((SC.WIDTH - 1) / 2, (SC.HEIGHT - 1) / 2)


@pytest.mark.parametrize("model,flags,aspect", [(0, SC.FLAGS_EXAMPLE, 1.0), (1, SC.FLAGS_FISHEYE, 0.0)])
def test_calibrate_recovers_ground_truth(model, flags, aspect):
    obj, img, rv, tv = SC.make_views(8, model, seed=6)
    intr, rvs, tvs, rms, it = CO.calibrate(model, obj, img, SC.WIDTH, SC.HEIGHT, flags, aspect)
    gt = SC.GT_PINHOLE if model == 0 else SC.GT_FISHEYE
    assert rms < 1e-9 and np.allclose(intr, gt, rtol=1e-8, atol=1e-8)
    assert np.allclose(rvs, rv, atol=1e-8) and np.allclose(tvs, tv, atol=1e-6)


def test_calibrate_reaches_the_least_squares_minimum():
    """Noisy views: the CvLevMarq restatement ends at the minimum scipy's trust-region solver finds."""
    obj, img, rv, tv = SC.make_views(8, 0, seed=7, noise_px=0.2)
    flags, aspect = SC.FLAGS_EXAMPLE, 1.0
    intr, rvs, tvs, rms, it = CO.calibrate(0, obj, img, SC.WIDTH, SC.HEIGHT, flags, aspect)
    free = np.concatenate([CO.free_mask(0, flags), np.ones(6 * 8)]).astype(bool)
    p0 = np.concatenate([intr, np.c_[rvs, tvs].ravel()])

    def fun(z):
        p = p0.copy()
        p[free] = z
        return CO.residuals(0, flags, aspect, p, obj, img)
    sol = least_squares(fun, p0[free] * (1 + 1e-3), method="trf", xtol=1e-14, ftol=1e-14, gtol=1e-14)
    rms_ref = np.sqrt((sol.fun ** 2).sum() / (8 * obj.shape[0]))
    assert abs(rms - rms_ref) < 1e-9
    # free intrinsics are fy, k1, k2, k3: the valley along k2/k3 is flat at this noise level, fy and k1 are not
    assert np.allclose(sol.x[:2], p0[free][:2], rtol=1e-4)


def test_pnp_consensus_drops_corrupted_points():
    obj, img, rv, tv = SC.make_views(3, 0, seed=8)
    bad = img[1].copy()
    bad[5] += (10, -8)
    r, t, inl = CO.pnp_consensus(0, SC.GT_PINHOLE, obj, bad)
    assert not inl[5] and inl.sum() == 35
    assert np.allclose(r, rv[1], atol=5e-5) and np.allclose(t, tv[1], atol=2e-3)
