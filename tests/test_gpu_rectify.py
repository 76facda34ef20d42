"""GPU: ecal_rectify_batch_dev vs oracle/rectify_oracle.cpp (CirclesEventFrame::rectifyFeatures,
event_camera_calib/src/CirclesEventFrame.cpp:417-638) on the detection pipeline's own outputs.
Bar: validity flags, frame verdicts and the rectified circles bit-identical (event pixels are integers, so the
nine fit sums are exact whatever their order; everything else is the same f64/f32 arithmetic without FMA)."""
import numpy as np
import pytest

import oracle_lib as O
import synth_rectify as SR
import synth_stream as SS

pytestmark = pytest.mark.gpu


def _run(fit_circle, fisheye=False):
    import torch
    import eventcalib_amd
    from eventcalib_amd.capi import RectifyParams
    from eventcalib_amd.pipeline import DetectPipeline
    ctx = eventcalib_amd.Context(0)
    pipe = DetectPipeline(ctx)
    SS.CAMERA = "fisheye" if fisheye else "pinhole"
    try:
        buf = SS.make_stream(400_000, rate=4.0e6, device="cpu", seed=4)
    finally:
        SS.CAMERA = "pinhole"
    dist = (SS.KB[0], SS.KB[1], SS.KB[2], SS.KB[3], 0.0) if fisheye else SR.DIST
    t, _, _ = SS.unpack_records(buf)
    t0, t1 = SS.tiled_windows(float(t[0]), float(t[-1]))
    pipe.set_windows(t0, t1)
    pipe.run(buf.cuda())
    torch.cuda.synchronize()
    S = len(t0)
    rng = np.random.default_rng(11)
    frames = np.arange(S, dtype=np.int32)[::-1].copy()           # keyframes need not be in window order
    tm = (np.asarray(t0) + np.asarray(t1))[frames] / 2
    # a third exact poses, a third slightly off (mixed outcomes per circle), a third clearly wrong
    scale = np.where(np.arange(S) % 3 == 0, 0.0, np.where(np.arange(S) % 3 == 1, 0.6, 3.0))
    shift = rng.normal(size=(S, 3)) * scale[:, None]
    pose = SR.poses_cw(tm, shift)
    lm = SR.landmarks_f32()
    prm = RectifyParams()
    prm.fx, prm.fy, prm.cx, prm.cy = SR.CAMERA
    for i, v in enumerate(dist):
        prm.dist[i] = v
    prm.model = 1 if fisheye else 0
    prm.width, prm.height = SS.SENSOR_W, SS.SENSOR_H
    prm.rows, prm.cols, prm.asymmetric = 9, 4, 1
    prm.circle_radius, prm.fit_circle = SS.RADIUS, int(fit_circle)
    d_frames = torch.tensor(frames).cuda()
    d_pose = torch.tensor(pose).cuda()
    d_lm = torch.tensor(lm).cuda()
    feat = torch.empty(S, 36, 3, dtype=torch.float64, device="cuda")
    valid = torch.empty(S, 36, dtype=torch.int32, device="cuda")
    info = torch.empty(S, 2, dtype=torch.int32, device="cuda")
    ctx.rectify_batch_dev(pipe.xy.data_ptr(), pipe.seg_off.data_ptr(), pipe.seg_cnt.data_ptr(),
                          pipe.kept_labels.data_ptr(), pipe.win_info.data_ptr(), d_frames.data_ptr(), d_pose.data_ptr(),
                          S, d_lm.data_ptr(), prm, feat.data_ptr(), valid.data_ptr(), info.data_ptr(),
                          torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    feat, valid, info = feat.cpu().numpy(), valid.cpu().numpy(), info.cpu().numpy()
    xy = pipe.xy.cpu().numpy()
    kept = pipe.kept_labels.cpu().numpy()
    off = pipe.seg_off[:2 * S].cpu().numpy()
    cnt = pipe.seg_cnt[:2 * S].cpu().numpy()
    n_ok = n_mixed = 0
    for f in range(S):
        s = frames[f]
        pos, neg = xy[off[2 * s]: off[2 * s] + cnt[2 * s]], xy[off[2 * s + 1]: off[2 * s + 1] + cnt[2 * s + 1]]
        kp, kn = kept[off[2 * s]: off[2 * s] + cnt[2 * s]], kept[off[2 * s + 1]: off[2 * s + 1] + cnt[2 * s + 1]]
        o_feat, o_valid, o_ok, o_erased = O.rectify(pos, neg, kp, kn, pose[f], SR.CAMERA, dist, SS.SENSOR_W,
                                                    SS.SENSOR_H, lm, 9, 4, True, SS.RADIUS, fit_circle=fit_circle, model=int(fisheye))
        assert np.array_equal(valid[f], o_valid.astype(np.int32)), "frame %d validity" % f
        assert (info[f, 0], info[f, 1]) == (o_ok, o_erased), "frame %d verdict" % f
        assert np.array_equal(feat[f], o_feat, equal_nan=True), "frame %d circles" % f
        n_ok += o_ok
        n_mixed += int(0 < o_erased < 36)
    assert n_ok >= S // 4 and n_mixed >= S // 6          # the cases really cover kept, partially erased and rejected frames
    ctx.close()


def test_rectify_matches_oracle():
    _run(False)


def test_rectify_fit_circle_mode_matches_oracle():
    _run(True)


def test_rectify_with_the_fisheye_projection_matches_oracle():
    """BASELINE configs[4]: the projections of rectifyFeatures through cv::fisheye::projectPoints (Kannala-Brandt stream)."""
    _run(False, fisheye=True)
