"""GPU: the calibration chain on one synthetic stream — packed events -> association against keyframe
circles -> continuous-time solve -> intrinsics (BASELINE config 3 at test size).  The keyframe circles
come from the ground truth (the init stage that would produce them — grid ordering, calibrateCamera,
PnP, rectifyFeatures — is a "next" row of SURVEY §8f)."""
import numpy as np
import pytest

import synth_solver_torch as ST
import synth_stream as SS

pytestmark = pytest.mark.gpu


def test_stream_to_intrinsics():
    import torch
    from scipy.spatial.transform import Rotation
    import eventcalib_amd
    from eventcalib_amd.capi import Solver, inverse_radial_distortion
    ctx = eventcalib_amd.Context(0)
    n, rate, t_start = 600_000, 1.0e6, 5.0
    buf = SS.make_stream(n, rate=rate, t_start=t_start, device="cpu", seed=21)
    t_first, t_last = t_start, t_start + (n - 1) / rate
    # keyframes every 4 ms, circles = projected ground-truth centres with their pixel radius
    kt = np.arange(t_first + 2e-3, t_last, 4e-3)
    R, C = SS.pose(torch.tensor(kt))
    lm = SS.landmarks()
    circ = np.zeros((len(kt), 36, 3))
    for i in range(36):
        c = SS.project(lm[i][None, :].expand(len(kt), 3), R, C).numpy()
        rim = [SS.project((lm[i] + SS.RADIUS * torch.tensor([np.cos(a), np.sin(a), 0.0]))[None, :].expand(len(kt), 3), R, C).numpy()
               for a in np.linspace(0, 2 * np.pi, 8, endpoint=False)]
        circ[:, i, :2] = c
        circ[:, i, 2] = np.mean([np.linalg.norm(r - c, axis=1) for r in rim], axis=0)
    d_ev = buf.cuda()
    d_kt, d_ci = torch.tensor(kt).cuda(), torch.tensor(circ).cuda()
    obs = torch.empty(n, 2, dtype=torch.float64, device="cuda")
    tm = torch.empty(n, dtype=torch.float64, device="cuda")
    lmid = torch.empty(n, dtype=torch.int32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    step = 5e-4
    ctx.associate_dev(d_ev.data_ptr(), n, d_kt.data_ptr(), d_ci.data_ptr(), len(kt), 36, t_first, t_last, 5 * step, 5.0,
                      obs.data_ptr(), tm.data_ptr(), lmid.data_ptr(), cnt.data_ptr(), 0)
    torch.cuda.synchronize()
    m = int(cnt.item())
    assert 0.80 * n < m < 0.97 * n          # ~90 % edge events; noise mostly fails the 5 px rim gate
    # spline layout as EventCalibSpline.cpp:63-85: range padded, one control point per 50 steps
    n_cp = max(4, int((t_last - t_first) / (50 * step)))
    knots = ST.uniform_knots(n_cp, t_first, t_last)
    grev = np.array([knots[i + 1:i + 4].mean() for i in range(n_cp)])
    Rg, Cg = SS.pose(torch.tensor(grev))
    q = Rotation.from_matrix(Rg.numpy()).as_quat()
    for i in range(1, n_cp):
        if q[i] @ q[i - 1] < 0:
            q[i] = -q[i]
    rng = np.random.default_rng(1)
    # start: OpenCV-style initial intrinsics 2 % off, inverse radial polynomial from the forward coefficients
    b5 = inverse_radial_distortion([SS.K1, SS.K2, SS.K3, 0.0])
    x0 = np.concatenate([[SS.FX * 1.02, SS.FY * 0.98, SS.CX + 2.0, SS.CY - 2.0], b5, q.ravel(),
                         (Cg.numpy() + 0.2 * rng.normal(size=(n_cp, 3))).ravel()])
    prob = dict(seg_cp_off=np.array([0, n_cp], np.uint32), knots=knots, obs=obs[:m].cpu().numpy(), time=tm[:m].cpu().numpy(),
                lm_id=lmid[:m].cpu().numpy().astype(np.uint32), seg_id=None, landmarks=lm.numpy(), circle_radius=SS.RADIUS,
                huber_a=0.2 * SS.RADIUS)
    s = Solver(ctx, prob)
    x, summ = s.solve(x0)
    assert summ.final_cost < 0.2 * summ.initial_cost
    # tolerance: focal lengths and principal point within 1 % / 1.5 px of the generating camera
    # (integer-pixel events, 10 % noise events, 0.6 s of motion, inverse-polynomial distortion model)
    assert abs(x[0] / SS.FX - 1) < 0.01 and abs(x[1] / SS.FY - 1) < 0.01, x[:4]
    assert abs(x[2] - SS.CX) < 1.5 and abs(x[3] - SS.CY) < 1.5, x[:4]
    rms = np.sqrt(2 * summ.final_cost / m)
    assert rms < 0.25        # cm on the board; one pixel is ~0.18 cm at 66 cm
    s.close()
    ctx.close()
