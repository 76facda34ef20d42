"""Init calibration on the GPU (ecal_calibrate_views, ecal_calib_view_blocks_dev, ecal_pnp_batch_dev) against the
numpy oracle (oracle/calib_oracle.py) and synthetic ground truth.  Floating point: tolerances stated per test."""
import numpy as np
import pytest

import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import synth_calib as SC  # noqa: E402
import calib_oracle as CO  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import eventcalib_amd
    c = eventcalib_amd.Context(0)
    yield c
    c.close()


def _dev(a):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a), device="cuda:0")


@pytest.mark.parametrize("model,flags,aspect", [(0, SC.FLAGS_EXAMPLE, 1.0), (0, 0, 0.0), (0, CO.FIX_K6 | CO.FIX_K5, 0.0),
                                                (1, SC.FLAGS_FISHEYE, 0.0), (1, 0, 0.0)])
def test_view_blocks_match_finite_differences(ctx, model, flags, aspect):
    """Analytic per-view blocks == J^T J / J^T r of the oracle's central-difference Jacobian (rel. 2e-6 of the block
    scale: the finite differences carry ~1e-7), cost bit-close (1e-12)."""
    import torch
    from eventcalib_amd import capi
    obj, img, rv, tv = SC.make_views(5, model, seed=3, noise_px=0.3)
    intr = (SC.GT_PINHOLE if model == 0 else SC.GT_FISHEYE).copy()
    if model == 0:
        intr[6:8] = 1e-3, -2e-3            # tangential terms and the rational denominators get exercised too
        intr[9:12] = 0.01, -0.02, 0.005
    else:
        intr[4] = 0.01
    p = np.concatenate([intr] + [np.concatenate([rv[v], tv[v]]) for v in range(5)])
    p[12:] += 1e-3 * np.random.default_rng(0).normal(size=30)
    d_blocks = torch.zeros(5, capi.CALIB_BLOCK_DOUBLES, dtype=torch.float64, device="cuda:0")
    d_obj, d_img, d_intr, d_view = _dev(obj), _dev(img), _dev(p[:12]), _dev(p[12:])
    capi.calib_view_blocks_dev(ctx, d_obj.data_ptr(), obj.shape[0], d_img.data_ptr(), 5, model, flags, aspect,
                               d_intr.data_ptr(), d_view.data_ptr(), 1, d_blocks.data_ptr(),
                               torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    B = d_blocks.cpu().numpy()
    for v in range(5):
        Hii, Hiv, Hvv, gi, gv, cost = CO.view_blocks(model, flags, aspect, p, obj, img, v)
        for got, ref in ((B[v, :144].reshape(12, 12), Hii), (B[v, 144:216].reshape(12, 6), Hiv),
                         (B[v, 216:252].reshape(6, 6), Hvv), (B[v, 252:264], gi), (B[v, 264:270], gv)):
            # compare entry-wise relative to the geometric scale of the two diagonals involved
            assert np.allclose(got, ref, rtol=2e-5, atol=2e-6 * np.abs(ref).max() + 1e-9), (model, flags, v)
        assert abs(B[v, 270] - cost) <= 1e-12 * max(1.0, cost)
    # cost-only mode writes the same cost
    d2 = torch.zeros_like(d_blocks)
    capi.calib_view_blocks_dev(ctx, d_obj.data_ptr(), obj.shape[0], d_img.data_ptr(), 5, model, flags, aspect,
                               d_intr.data_ptr(), d_view.data_ptr(), 0, d2.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert (d2[:, 270] == d_blocks[:, 270]).all()


@pytest.mark.parametrize("model", [0, 1])
def test_pnp_recovers_pose_and_flags_outliers(ctx, model):
    """IPPE (+ consensus) recovers the generating poses (noise-free: rvec 5e-5, tvec 2e-3 — the pinhole undistortion
    stops after 5 fixed-point iterations as OpenCV's) and equals the oracle's IPPE; corrupted circles are flagged."""
    import torch
    from eventcalib_amd import capi
    F = 16
    gt = SC.GT_PINHOLE if model == 0 else SC.GT_FISHEYE
    obj, img, rv, tv = SC.make_views(F, model, seed=5)
    img = img.copy()
    img[3, 7] += (9.0, -7.0)           # two corrupted detections
    img[9, 30] += (-12.0, 5.0)
    d_pose = torch.zeros(F, 6, dtype=torch.float64, device="cuda:0")
    d_inl = torch.zeros(F, obj.shape[0], dtype=torch.int32, device="cuda:0")
    d_err = torch.zeros(F, dtype=torch.float64, device="cuda:0")
    d_ok = torch.zeros(F, dtype=torch.int32, device="cuda:0")
    d_obj, d_img, d_intr = _dev(obj), _dev(img), _dev(gt)
    capi.pnp_batch_dev(ctx, d_obj.data_ptr(), obj.shape[0], d_img.data_ptr(), None, F, model, d_intr.data_ptr(), 4.0, 3, 0,
                       d_pose.data_ptr(), d_inl.data_ptr(), d_err.data_ptr(), d_ok.data_ptr(),
                       torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    pose, inl = d_pose.cpu().numpy(), d_inl.cpu().numpy()
    assert d_ok.cpu().numpy().all()
    exp_inl = np.ones_like(inl)
    exp_inl[3, 7] = exp_inl[9, 30] = 0
    assert (inl == exp_inl).all()
    assert np.abs(pose[:, :3] - rv).max() < 5e-5 and np.abs(pose[:, 3:] - tv).max() < 2e-3
    for f in (0, 3, 9):
        r_o, t_o, inl_o = CO.pnp_consensus(model, gt, obj, img[f])
        assert (inl_o == inl[f].astype(bool)).all()
        assert np.abs(pose[f, :3] - r_o).max() < 1e-9 and np.abs(pose[f, 3:] - t_o).max() < 1e-8
    # with refinement the ML pose of the clean frames is exact up to the FLT_EPSILON parameter-change stop
    capi.pnp_batch_dev(ctx, d_obj.data_ptr(), obj.shape[0], d_img.data_ptr(), None, F, model, d_intr.data_ptr(), 4.0, 3, 20,
                       d_pose.data_ptr(), d_inl.data_ptr(), d_err.data_ptr(), d_ok.data_ptr(),
                       torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    pose = d_pose.cpu().numpy()
    assert np.abs(pose[:, :3] - rv).max() < 1e-7 and np.abs(pose[:, 3:] - tv).max() < 1e-5


@pytest.mark.parametrize("model,flags,aspect,V", [(0, SC.FLAGS_EXAMPLE, 1.0, 24), (0, CO.FIX_K4 | CO.FIX_K5 | CO.FIX_K6, 0.0, 24),
                                                  (1, SC.FLAGS_FISHEYE, 0.0, 16)])
def test_calibrate_recovers_ground_truth(ctx, model, flags, aspect, V):
    """Noise-free views: the generating intrinsics and poses come back (1e-7 relative), rms ~ 0."""
    from eventcalib_amd import capi
    obj, img, rv, tv = SC.make_views(V, model, seed=11)
    out = capi.calibrate_views(ctx, obj, img, SC.WIDTH, SC.HEIGHT, model, flags, aspect)
    gt = SC.GT_PINHOLE if model == 0 else SC.GT_FISHEYE
    assert out["rms"] < 1e-8
    assert np.abs(out["intr"][:4] / gt[:4] - 1).max() < 1e-7
    assert np.abs(out["intr"][4:] - gt[4:]).max() < 1e-6
    assert np.abs(out["rvecs"] - rv).max() < 1e-7 and np.abs(out["tvecs"] - tv).max() < 1e-5


@pytest.mark.parametrize("model,flags,aspect", [(0, SC.FLAGS_EXAMPLE, 1.0), (1, SC.FLAGS_FISHEYE, 0.0)])
def test_calibrate_matches_oracle_on_noisy_views(ctx, model, flags, aspect):
    """0.2 px noise: same minimum as the oracle's dense restatement of the OpenCV loop (the GPU path eliminates the
    views by Schur complements, the oracle factorises the full matrix): intrinsics 1e-6 relative, rms 1e-9."""
    from eventcalib_amd import capi
    obj, img, rv, tv = SC.make_views(10, model, seed=21, noise_px=0.2)
    out = capi.calibrate_views(ctx, obj, img, SC.WIDTH, SC.HEIGHT, model, flags, aspect)
    intr, rvs, tvs, rms, it = CO.calibrate(model, obj, img, SC.WIDTH, SC.HEIGHT, flags, aspect)
    assert abs(out["rms"] - rms) < 1e-7
    assert np.abs(out["intr"][:4] / intr[:4] - 1).max() < 1e-5
    assert np.abs(out["intr"][4:] - intr[4:]).max() < 1e-4
    assert np.abs(out["per_view_err"]).max() < 1.0


def test_calibrate_rejects_bad_input(ctx):
    from eventcalib_amd import capi
    obj, img, _, _ = SC.make_views(4, 0, seed=1)
    bad = obj.copy()
    bad[0, 2] = 1.0
    with pytest.raises(capi.EcalError):
        capi.calibrate_views(ctx, bad, img, SC.WIDTH, SC.HEIGHT)
    with pytest.raises(capi.EcalError):
        capi.calibrate_views(ctx, obj, img[:0], SC.WIDTH, SC.HEIGHT)
