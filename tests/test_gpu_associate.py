"""GPU parity: event -> residual association vs the oracle (bit-exact: same arithmetic, index outputs)."""
import numpy as np
import pytest

import oracle_lib as O
import synth_stream as SS

pytestmark = pytest.mark.gpu


def _keyframes(t_first, t_last, torch):
    """Keyframes every 4 ms with the projected ground-truth circles of the synthetic stream."""
    kt = np.arange(t_first + 2e-3, t_last, 4e-3)
    tt = torch.tensor(kt, dtype=torch.float64)
    R, C = SS.pose(tt)
    lm = SS.landmarks()
    circ = np.zeros((len(kt), 36, 3))
    for i in range(36):
        c = SS.project(lm[i][None, :].expand(len(kt), 3), R, C).numpy()
        e = SS.project((lm[i] + torch.tensor([SS.RADIUS, 0.0, 0.0]))[None, :].expand(len(kt), 3), R, C).numpy()
        circ[:, i, :2] = c
        circ[:, i, 2] = np.linalg.norm(e - c, axis=1)
    return kt, circ


def test_associate_matches_oracle():
    import torch
    import eventcalib_amd
    ctx = eventcalib_amd.Context(0)
    buf = SS.make_stream(120000, device="cpu", seed=8)
    t, _, _ = SS.unpack_records(buf)
    t_first, t_last = float(t[0]), float(t[-1])
    kt, circ = _keyframes(t_first, t_last, torch)
    keep = (np.arange(len(kt)) % 7) != 3              # drop some keyframes: events too far from any are rejected
    kt, circ = kt[keep], circ[keep]
    d_ev = buf.cuda()
    d_kt = torch.tensor(kt).cuda()
    d_ci = torch.tensor(circ).cuda()
    n = 120000
    obs = torch.empty(n, 2, dtype=torch.float64, device="cuda")
    tm = torch.empty(n, dtype=torch.float64, device="cuda")
    lm = torch.empty(n, dtype=torch.int32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    for (t_min, t_max) in ((t_first, t_last), (t_first + 0.01, t_last - 0.02)):
        ctx.associate_dev(d_ev.data_ptr(), n, d_kt.data_ptr(), d_ci.data_ptr(), len(kt), 36, t_min, t_max, 2.5e-3, 5.0,
                          obs.data_ptr(), tm.data_ptr(), lm.data_ptr(), cnt.data_ptr(), 0)
        torch.cuda.synchronize()
        m = int(cnt.item())
        oo, ot, ol = O.associate(buf.numpy(), kt, circ, t_min, t_max, 2.5e-3, 5.0)
        assert m == len(ot) and m > 10000
        assert np.array_equal(obs[:m].cpu().numpy(), oo)
        assert np.array_equal(tm[:m].cpu().numpy(), ot)
        assert np.array_equal(lm[:m].cpu().numpy().astype(np.uint32), ol)
    # no keyframes -> nothing associated
    ctx.associate_dev(d_ev.data_ptr(), n, 0, 0, 0, 36, t_first, t_last, 2.5e-3, 5.0, obs.data_ptr(), tm.data_ptr(),
                      lm.data_ptr(), cnt.data_ptr(), 0)
    torch.cuda.synchronize()
    assert int(cnt.item()) == 0
    ctx.close()


def test_all_segments_in_one_pass_and_the_solver_built_in_place():
    """ecal_associate_ranges_dev (every spline segment's range in ONE pass, segment ids out, count left on the device) ==
    the oracle's association range by range; ecal_solver_create_dev on those device arrays == ecal_solver_create on the
    downloaded ones (same chunk table, same normal equations bit for bit)."""
    import torch
    import eventcalib_amd
    from eventcalib_amd import capi
    import synth_solver as SV
    ctx = eventcalib_amd.Context(0)
    try:
        n = 200000
        buf = SS.make_stream(n, device="cpu", seed=9)
        t, _, _ = SS.unpack_records(buf)
        t_first, t_last = float(t[0]), float(t[-1])
        kt, circ = _keyframes(t_first, t_last, torch)
        span = t_last - t_first
        ranges = np.array([[t_first + 0.01 * span, t_first + 0.30 * span], [t_first + 0.33 * span, t_first + 0.34 * span],
                           [t_first + 0.50 * span, t_first + 0.50 * span],            # (one instant: empty)
                           [t_first + 0.55 * span, t_first + 0.99 * span]])
        d_ev, d_kt, d_ci, d_rg = buf.cuda(), torch.tensor(kt).cuda(), torch.tensor(circ).cuda(), torch.tensor(ranges).cuda()
        obs = torch.empty(n, 2, dtype=torch.float64, device="cuda")
        tm = torch.empty(n, dtype=torch.float64, device="cuda")
        lm = torch.empty(n, dtype=torch.int32, device="cuda")
        sg = torch.empty(n, dtype=torch.int32, device="cuda")
        cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
        ctx.associate_ranges_dev(d_ev.data_ptr(), n, d_kt.data_ptr(), d_ci.data_ptr(), len(kt), 36, d_rg.data_ptr(), len(ranges),
                                 2.5e-3, 5.0, obs.data_ptr(), tm.data_ptr(), lm.data_ptr(), sg.data_ptr(), cnt.data_ptr(), 0)
        torch.cuda.synchronize()
        m = int(cnt.item())
        parts = [O.associate(buf.numpy(), kt, circ, a, b, 2.5e-3, 5.0) for a, b in ranges]
        assert m == sum(len(p[1]) for p in parts) and m > 50000
        assert np.array_equal(obs[:m].cpu().numpy(), np.concatenate([p[0] for p in parts]))
        assert np.array_equal(tm[:m].cpu().numpy(), np.concatenate([p[1] for p in parts]))
        assert np.array_equal(lm[:m].cpu().numpy().astype(np.uint32), np.concatenate([p[2] for p in parts]))
        assert np.array_equal(sg[:m].cpu().numpy(), np.concatenate([np.full(len(p[1]), r) for r, p in enumerate(parts)]))
        # the solver on the device arrays: four segments with clamped knot vectors over their ranges
        rng = np.random.default_rng(2)
        ncp = [9, 4, 4, 12]
        seg_cp_off = np.concatenate([[0], np.cumsum(ncp)]).astype(np.uint32)
        knots = []
        for (a, b), c in zip(ranges, ncp):
            b = max(b, a + 1e-6)
            inner = np.linspace(a, b, c - 2)
            knots.append(np.concatenate([[a] * 3, inner, [b] * 3]))
        prob = dict(seg_cp_off=seg_cp_off, knots=np.concatenate(knots), landmarks=SS.landmarks().numpy().astype(np.float64),
                    circle_radius=1.75, huber_a=0.35)
        dev_solver = capi.Solver(ctx, prob, device_arrays=(obs.data_ptr(), tm.data_ptr(), lm.data_ptr(), sg.data_ptr(), n, cnt.data_ptr()))
        host_solver = capi.Solver(ctx, dict(prob, obs=obs[:m].cpu().numpy(), time=tm[:m].cpu().numpy(),
                                            lm_id=lm[:m].cpu().numpy().astype(np.uint32), seg_id=sg[:m].cpu().numpy().astype(np.uint32)))
        assert dev_solver.n_res == host_solver.n_res == m and dev_solver.n_chunks == host_solver.n_chunks
        n_cp = int(seg_cp_off[-1])
        q = rng.normal(size=(n_cp, 4)) * 0.05 + np.array([0.0, 0.0, 0.0, 1.0])
        x = np.concatenate([SV.GT_INTR, (q / np.linalg.norm(q, axis=1, keepdims=True)).ravel(),
                            (rng.normal(size=(n_cp, 3)) * 5 + np.array([0.0, 0.0, -60.0])).ravel()])
        a, b = dev_solver.evaluate(x), host_solver.evaluate(x)
        assert np.isfinite(a).all() and np.abs(a).max() > 0
        assert np.allclose(a, b, rtol=1e-12, atol=1e-9 * np.abs(b).max())     # (atomic accumulation order differs run to run)
        # wrong input is refused as by the host form: a time outside its segment's knot range
        tm2 = tm.clone()
        tm2[5] = t_first - 1.0
        with pytest.raises(capi.EcalError):
            capi.Solver(ctx, prob, device_arrays=(obs.data_ptr(), tm2.data_ptr(), lm.data_ptr(), sg.data_ptr(), n, cnt.data_ptr()))
        dev_solver.close()
        host_solver.close()
    finally:
        ctx.close()
