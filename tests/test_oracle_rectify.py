"""CPU: behaviour of the rectifyFeatures restatement (oracle/rectify_oracle.cpp) on the synthetic stream.
The reference holds no test for this function (parity unpinned); what is checked here is that the restated
procedure does what CirclesEventFrame.cpp:417-638 describes on data with known ground truth."""
import numpy as np
import torch

import oracle_lib as O
import synth_rectify as SR
import synth_stream as SS


def _window(seed=3, n=6000, rate=4.0e6):
    buf = SS.make_stream(n, rate=rate, device="cpu", seed=seed)
    rec = buf.numpy()
    t, _, _ = SS.unpack_records(buf)
    t0, t1 = float(t[0]), float(t[-1])
    lo, hi = O.window_bounds(rec, t0, t1)
    pos, neg, _ = O.event_frame(rec, lo, hi, "reference")
    ex = O.extract_candidates(pos, neg, 4.0, 2, 5, 36, 15.511363636363637)
    return 0.5 * (t0 + t1), pos, neg, ex


def _truth(tm):
    R, C = SS.pose(torch.tensor([tm], dtype=torch.float64))
    lm = SS.landmarks()
    return np.stack([SS.project(lm[i][None], R, C).numpy()[0] for i in range(36)])


def test_true_pose_keeps_all_circles():
    tm, pos, neg, ex = _window()
    pose = SR.poses_cw([tm])[0]
    feat, valid, ok, erased = O.rectify(pos, neg, ex["kept_pos"], ex["kept_neg"], pose, SR.CAMERA, SR.DIST,
                                        SS.SENSOR_W, SS.SENSOR_H, SR.landmarks_f32(), 9, 4, True, SS.RADIUS)
    assert ok == 1 and erased == 0 and valid.all()
    gt = _truth(tm)
    assert np.abs(feat[:, :2] - gt).max() < 2.0                 # refit centres sit on the projected centres
    assert (feat[:, 2] > 5).all() and (feat[:, 2] < 14).all()


def test_shifted_pose_erases_and_rejects():
    tm, pos, neg, ex = _window()
    pose = SR.poses_cw([tm], shift=np.array([[3.0, 0.0, 0.0]]))[0]   # 3 cm sideways: ~16 px in the image
    feat, valid, ok, erased = O.rectify(pos, neg, ex["kept_pos"], ex["kept_neg"], pose, SR.CAMERA, SR.DIST,
                                        SS.SENSOR_W, SS.SENSOR_H, SR.landmarks_f32(), 9, 4, True, SS.RADIUS)
    assert ok == 0 and erased >= 8
    assert np.isnan(feat[valid == 0]).all()


def test_border_score_and_twenty_percent_rule():
    tm, pos, neg, ex = _window()
    pose = SR.poses_cw([tm])[0]
    lm = SR.landmarks_f32()
    # take the events of the first pattern row away: its 4 circles are erased -> border score fails the frame
    gt = _truth(tm)
    def far(p):
        d = np.linalg.norm(p[:, None, :] - gt[None, :4, :], axis=2).min(axis=1)
        return d > 16
    kp, kn = ex["kept_pos"].copy(), ex["kept_neg"].copy()
    kp[~far(pos)] = -1
    kn[~far(neg)] = -1
    feat, valid, ok, erased = O.rectify(pos, neg, kp, kn, pose, SR.CAMERA, SR.DIST, SS.SENSOR_W, SS.SENSOR_H, lm, 9, 4,
                                        True, SS.RADIUS)
    assert (valid[:4] == 0).all() and erased >= 4 and ok == 0
    # with fitCircle the border test is skipped; 4 of 36 erased is below the 20 % rule (>= 7.2)
    feat, valid, ok2, erased2 = O.rectify(pos, neg, kp, kn, pose, SR.CAMERA, SR.DIST, SS.SENSOR_W, SS.SENSOR_H, lm, 9, 4,
                                          True, SS.RADIUS, fit_circle=True)
    assert erased2 == erased and ok2 == (1 if erased < 8 else 0)
