"""GPU, BASELINE configs[2] (50 M events, the full event_camera_calib chain, pinhole + radial): what bench.py's end_to_end leg
prints, ASSERTED — keyframe search with the reference's gate at the reference's piece count for a 256-thread host
(eventCameraCalib.cpp:172-173), init calibration, PnP / checkPose / rectifyFeatures, spline fit, association of every event,
the continuous-time solve.  Bars against the generating camera (tolerances as tests/test_gpu_init_chain.py's, stated there):
refined fx, fy within 1e-3, principal point within 0.3 px of its floored position, the refined undistortion map within 0.5 px
of the generating radial model over the rays within 0.30 of the optical axis (108 px around the principal point), at
most 10 LM iterations.  The chain's stage-by-stage parity with the CPU oracle chain is tests/test_gpu_oracle_chain.py (2 M events);
the keyframe search's parity with the sequential oracle at this piece count is tests/test_gpu_adaptive.py (12.7 M events)."""
import numpy as np
import pytest

import synth_stream as SS

pytestmark = pytest.mark.gpu
N_EVENTS = 50_000_000
PIECES = 1270


def test_the_50M_event_chain_recovers_the_generating_camera():
    import torch
    import eventcalib_amd
    from eventcalib_amd.calibrate import calibrate_stream
    if torch.cuda.get_device_properties(0).total_memory < 40e9:
        pytest.skip("needs ~10 GB of device memory")
    SS.TRAJECTORY = "orbit"
    try:
        ev = SS.make_stream(N_EVENTS, rate=1.0e6, t_start=5.0, device="cuda", seed=21)
        ctx = eventcalib_amd.Context(0)
        try:
            r = calibrate_stream(ctx, ev, 5.0, 5.0 + (N_EVENTS - 1) / 1e6, piece_num=PIECES)
            t = torch.tensor(r["trajectory"][:, 0])
            R, C = SS.pose(t)
        finally:
            ctx.close()
    finally:
        SS.TRAJECTORY = "hover"
    fx, fy, cx, cy = r["intrinsics"][:4]
    und = SS.undistortion_error_px(r["intrinsics"])
    print("\n[chain 50 M] keyframes %d, accepted %d, residuals %d, LM iterations %d, fx err %.2e, fy err %.2e, cx %+.3f px, cy %+.3f px, "
          "undistortion map max %.3f px, trajectory max %.3f cm" % (
              r["keyframes"], r["init"]["accepted"], r["spline"]["residuals"], r["spline"]["iterations"], fx / SS.FX - 1, fy / SS.FY - 1,
              cx - (SS.CX - 0.5), cy - (SS.CY - 0.5), und, np.abs(r["trajectory"][:, 1:4] - C.numpy()).max()))
    assert r["keyframes"] >= 8000 and r["init"]["accepted"] >= 5000 and r["init"]["views"] == 200
    assert r["spline"]["residuals"] >= 25_000_000 and r["spline"]["iterations"] <= 10
    assert r["spline"]["final_cost"] < r["spline"]["initial_cost"]
    assert abs(fx / SS.FX - 1) < 1e-3 and abs(fy / SS.FY - 1) < 1e-3
    assert abs(cx - (SS.CX - 0.5)) < 0.3 and abs(cy - (SS.CY - 0.5)) < 0.3
    assert und < 0.5
    assert np.abs(r["trajectory"][:, 1:4] - C.numpy()).max() < 0.3
