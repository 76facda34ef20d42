"""GPU parity: residual / Jacobian / normal equations and the LM solve vs the oracle.
Floating point (f64): tolerances are stated per assertion."""
import numpy as np
import pytest

import oracle_lib as O
import ref_lm
import synth_solver as SV

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import eventcalib_amd
    c = eventcalib_amd.Context(0)
    yield c
    c.close()


def _dense(acc, n_cp):
    from eventcalib_amd.capi import unpack_normal
    return unpack_normal(acc, n_cp)


@pytest.mark.parametrize("n_res,n_cp", [(500, 6), (3000, 9), (40000, 4)])
def test_normal_equations_match_oracle(ctx, n_res, n_cp):
    from eventcalib_amd.capi import Solver
    rng = np.random.default_rng(n_res)
    prob, x = SV.make_problem(n_res, n_cp=n_cp, seed=n_res, pixel_noise=0.5)
    y = SV.perturb(x, n_cp, rng, intr_rel=0.01, rot=0.005, trans=0.2)   # large enough to put residuals on the Huber tail
    s = Solver(ctx, prob)
    acc = s.evaluate(y, True)
    cost, g, H = _dense(acc, n_cp)
    oc, og, oH = O.solver_evaluate(prob, y)
    # f64 sums of ~n_res terms in a different order (atomics) and FMA contraction: relative 1e-11
    assert abs(cost - oc) <= 1e-11 * abs(oc)
    assert np.abs(g - og).max() <= 1e-10 * np.abs(og).max()
    assert np.abs(H - oH).max() <= 1e-10 * np.abs(oH).max()
    assert abs(s.evaluate(y, False)[0] - oc) <= 1e-11 * abs(oc)
    if n_res == 40000:
        assert s.n_chunks > n_cp - 3      # a span with more than 16384 residuals is split into several chunks
    s.close()


@pytest.mark.parametrize("n_res,n_cp", [(600, 6), (20000, 5)])
def test_so3_normal_equations_match_oracle(ctx, n_res, n_cp):
    """useSO3 = 1: cumulative SO3 spline + LocalParameterizationSO3 (EventCalibSpline.hpp:65-135, BsplineSO3.hpp:190-221)."""
    from eventcalib_amd.capi import Solver
    rng = np.random.default_rng(n_res + 1)
    prob, x = SV.make_problem(n_res, n_cp=n_cp, seed=n_res + 1, pixel_noise=0.5, use_so3=True)
    y = SV.perturb(x, n_cp, rng, intr_rel=0.01, rot=0.005, trans=0.2)
    s = Solver(ctx, prob)
    acc = s.evaluate(y, True)
    cost, g, H = _dense(acc, n_cp)
    oc, og, oH = O.solver_evaluate(prob, y)
    assert abs(cost - oc) <= 1e-11 * abs(oc)
    assert np.abs(g - og).max() <= 1e-9 * np.abs(og).max()
    assert np.abs(H - oH).max() <= 1e-9 * np.abs(oH).max()
    # and it is a different function from the quaternion-blend variant
    qc = O.solver_evaluate(dict(prob, use_so3=False), y, want_H=False)[0]
    assert abs(qc - oc) > 1e-9 * abs(oc)
    s.close()


def test_so3_lm_recovers_ground_truth(ctx):
    from eventcalib_amd.capi import Solver
    rng = np.random.default_rng(14)
    n_cp = 8
    prob, x_gt = SV.make_problem(4000, n_cp=n_cp, seed=14, use_so3=True)
    x0 = SV.perturb(x_gt, n_cp, rng)
    s = Solver(ctx, prob)
    x, summ = s.solve(x0)
    assert summ.termination == 0 and summ.final_cost < 1e-12 * summ.initial_cost + 1e-16
    assert np.abs(x[:4] / x_gt[:4] - 1).max() < 1e-6
    assert np.abs(x[4:9] - x_gt[4:9]).max() < 1e-5
    assert np.abs(np.linalg.norm(x[9:9 + 4 * n_cp].reshape(n_cp, 4), axis=1) - 1).max() < 1e-12   # stays on the manifold
    xr, hist, it = ref_lm.solve(prob, x0)
    assert np.abs(x[:9] - xr[:9]).max() <= 1e-7 * np.abs(xr[:9]).max()
    s.close()


def test_two_segments(ctx):
    from eventcalib_amd.capi import Solver
    prob, x = SV.make_problem(1200, n_cp=6, seed=9, n_segments=2, pixel_noise=0.2)
    s = Solver(ctx, prob)
    acc = s.evaluate(x, True)
    cost, g, H = _dense(acc, 12)
    # the oracle handles one segment: evaluate each segment separately (shared intrinsics) and add
    tot_c, tot_g, tot_H = 0.0, np.zeros_like(g), np.zeros_like(H)
    for seg in range(2):
        m = prob["seg_id"] == seg
        p1 = dict(prob, seg_cp_off=np.array([0, 6], np.uint32), knots=prob["knots"][10 * seg: 10 * seg + 10], obs=prob["obs"][m],
                  time=prob["time"][m], lm_id=prob["lm_id"][m], seg_id=None)
        x1 = np.concatenate([x[:9], x[9 + 24 * seg: 9 + 24 * seg + 24], x[9 + 48 + 18 * seg: 9 + 48 + 18 * seg + 18]])
        c1, g1, H1 = O.solver_evaluate(p1, x1)
        idx = np.concatenate([np.arange(9), 9 + 36 * seg + np.arange(36)])
        tot_c += c1
        tot_g[idx] += g1
        tot_H[np.ix_(idx, idx)] += H1
    assert abs(cost - tot_c) <= 1e-11 * abs(tot_c) + 1e-20
    assert np.abs(g - tot_g).max() <= 1e-10 * np.abs(tot_g).max()
    assert np.abs(H - tot_H).max() <= 1e-10 * np.abs(tot_H).max()
    # no coupling between the control points of different segments
    assert np.abs(H[9:45, 45:81]).max() == 0.0
    s.close()


@pytest.mark.parametrize("n_cp,n_res", [(8, 4000), (60, 30000)])
def test_lm_recovers_ground_truth_and_matches_reference_loop(ctx, n_cp, n_res, monkeypatch):
    from eventcalib_amd.capi import Solver
    rng = np.random.default_rng(4)
    prob, x_gt = SV.make_problem(n_res, n_cp=n_cp, seed=4)
    x0 = SV.perturb(x_gt, n_cp, rng)
    s = Solver(ctx, prob)
    x, summ = s.solve(x0)
    assert summ.termination == 0 and summ.final_cost < 1e-12 * summ.initial_cost + 1e-16
    # intrinsics back to the ground truth: relative 1e-6 (noise-free data; the gauge is fixed by the board)
    assert np.abs(x[:4] / x_gt[:4] - 1).max() < 1e-6
    assert np.abs(x[4:9] - x_gt[4:9]).max() < 1e-5
    # same minimiser as the dense numpy loop on the oracle: iterates agree closely
    xr, hist, it = ref_lm.solve(prob, x0)
    assert np.abs(x[:9] - xr[:9]).max() <= 1e-7 * np.abs(xr[:9]).max()
    assert abs(summ.successful_steps + 1 - len(hist)) <= 2
    s.close()


def test_lm_with_noise_and_outliers(ctx):
    from eventcalib_amd.capi import Solver
    rng = np.random.default_rng(6)
    n_cp = 7
    prob, x_gt = SV.make_problem(6000, n_cp=n_cp, seed=6, pixel_noise=0.4)
    bad = rng.random(6000) < 0.05
    prob["obs"][bad] += rng.normal(0, 15, size=(int(bad.sum()), 2))      # outliers -> Huber region
    x0 = SV.perturb(x_gt, n_cp, rng)
    s = Solver(ctx, prob)
    x, summ = s.solve(x0)
    xr, hist, it = ref_lm.solve(prob, x0)
    assert summ.final_cost <= hist[-1] * (1 + 1e-6)
    # with 0.4 px noise, 5 % outliers and half a second of small motion the focal length / board distance
    # direction is weakly observable, so closeness to the truth is loose (10 %); what is tight is the
    # agreement with the oracle's minimiser: same minimum to 1e-5 relative
    assert np.abs(x[:2] / x_gt[:2] - 1).max() < 1e-1
    assert np.abs(x[:9] - xr[:9]).max() <= 1e-5 * np.abs(xr[:9]).max()
    s.close()


def test_inverse_radial(ctx):
    from eventcalib_amd.capi import inverse_radial_distortion
    b = inverse_radial_distortion([-0.34991902, -0.014698517, 0.59684463, 0.0])
    assert np.array_equal(b, O.inverse_radial([-0.34991902, -0.014698517, 0.59684463, 0.0]))


@pytest.mark.parametrize("use_so3", [False, True])
def test_per_residual_rows_match_dual_numbers(ctx, use_so3):
    """ecal_residuals (the Ceres CostFunction::Evaluate seam, EventCalibSpline.hpp:137-146,231-240): raw residual and raw
    1 x 33 tangent row of EVERY residual against forward-mode dual numbers through the oracle's restated functor +
    local parameterisation (oracle_residual / oracle_residual_so3); spans and basis values against oracle_find_span /
    oracle_basis (BsplineReal.hpp:107-145,208-231)."""
    import ctypes
    from eventcalib_amd.capi import Solver
    L = O.lib()
    dp = ctypes.POINTER(ctypes.c_double)
    L.oracle_find_span.argtypes = [dp, ctypes.c_uint32, ctypes.c_double]
    L.oracle_find_span.restype = ctypes.c_uint32
    L.oracle_basis.argtypes = [dp, ctypes.c_uint32, ctypes.c_double, dp]
    fn = L.oracle_residual_so3 if use_so3 else L.oracle_residual
    fn.argtypes = [dp, dp, dp, dp, dp, dp, ctypes.c_double, dp, dp]
    fn.restype = ctypes.c_double
    n_res, n_cp = 2500, 7
    rng = np.random.default_rng(91)
    prob, x = SV.make_problem(n_res, n_cp=n_cp, seed=91, pixel_noise=0.5, use_so3=use_so3)
    y = SV.perturb(x, n_cp, rng, intr_rel=0.01, rot=0.005, trans=0.2)
    s = Solver(ctx, prob)
    r, J, cp0 = s.residuals(y)
    r_only, none, _ = s.residuals(y, with_jacobian=False)
    assert none is None and np.array_equal(r_only, r)
    intr = np.ascontiguousarray(y[:9])
    q = np.ascontiguousarray(y[9:9 + 4 * n_cp]).reshape(n_cp, 4)
    t = np.ascontiguousarray(y[9 + 4 * n_cp:]).reshape(n_cp, 3)
    kn = np.ascontiguousarray(prob["knots"], np.float64)
    lms = np.ascontiguousarray(prob["landmarks"], np.float64).reshape(-1, 3)
    p = lambda a: a.ctypes.data_as(dp)
    worst_r = worst_j = 0.0
    for k in range(0, n_res, 3):
        u = float(prob["time"][k])
        span = L.oracle_find_span(p(kn), n_cp, u)
        assert cp0[k] == span - 3
        b4 = np.zeros(4)
        L.oracle_basis(p(kn), span, u, p(b4))
        q4 = np.ascontiguousarray(q[span - 3:span + 1]).copy()
        t4 = np.ascontiguousarray(t[span - 3:span + 1]).copy()
        obs = np.ascontiguousarray(prob["obs"][k], np.float64).copy()
        lm = lms[int(prob["lm_id"][k])].copy()
        J33 = np.zeros(33)
        rr = fn(p(intr), p(q4), p(t4), p(b4), p(obs), p(lm), float(prob["circle_radius"]), None, p(J33))
        worst_r = max(worst_r, abs(r[k] - rr))
        worst_j = max(worst_j, np.abs(J[k] - J33).max() / max(1.0, np.abs(J33).max()))
    # f64, analytic derivative vs dual numbers: stated tolerance 1e-11 (absolute on r in cm, relative to the row's largest entry on J)
    assert worst_r < 1e-11 and worst_j < 1e-11, (worst_r, worst_j)
    s.close()


@pytest.mark.parametrize("n_cp,n_res", [(4, 2000), (9, 3000), (27, 6000), (28, 6000), (50, 12000), (131, 30000), (700, 150000)])
def test_partitioned_arrow_solves_match_the_sequential_solve(ctx, n_cp, n_res):
    """The host's partitioned banded-arrow Cholesky (arrow_host_parts.hpp: interiors in parallel, separators' reduced system,
    intrinsics, back substitution; the plain partition and the streamed evaluation's) against the sequential factorisation of
    the same scaled, damped system built from the GPU's normal equations: f64, different elimination order -> relative 1e-9 on
    the step; and the step solves the system (dense check)."""
    import ctypes
    from eventcalib_amd.capi import Solver
    rng = np.random.default_rng(n_cp)
    prob, x = SV.make_problem(n_res, n_cp=n_cp, seed=n_cp, pixel_noise=0.5)
    y = SV.perturb(x, n_cp, rng, intr_rel=0.01, rot=0.005, trans=0.2)
    s = Solver(ctx, prob)
    acc = np.ascontiguousarray(s.evaluate(y, True))
    nt = 6 * n_cp + 9
    L = ctx._L
    L.ecal_debug_arrow_solve.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                         ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    L.ecal_debug_arrow_solve.restype = ctypes.c_int
    from eventcalib_amd.capi import unpack_normal
    cost, g, H = unpack_normal(acc, n_cp)
    # unknown order of the dense form: [intr 9 | cp ...]; of the solver: [cp ... | intr 9]
    diag = np.concatenate([np.diag(H)[9:], np.diag(H)[:9]])
    scale = np.ascontiguousarray(1.0 / (1.0 + np.sqrt(diag)))
    for radius in (1e4, 3.0):
        host = np.zeros(nt)
        fail = ctypes.c_int(-1)
        rc = L.ecal_debug_arrow_solve(s._h, acc.ctypes.data, scale.ctypes.data, radius, 1e-6, 1e32, host.ctypes.data, ctypes.byref(fail), 0)
        assert rc == 0 and fail.value == 0, (rc, fail.value)
        assert np.abs(host).max() > 0
        # the host routine's own partitioned form (arrow_host_parts.hpp: interiors on several cores, separators' reduced system):
        # mode 2 of the hook, 4 interiors here, what ecal_solver_solve uses on long splines
        import os
        for parts in ([None] if n_cp >= 28 else []) + ([2, 3, 5, 8, 16] if n_cp >= 131 else []):
            if parts is None:
                os.environ.pop("ECAL_HOST_ARROW_PARTS", None)
                __import__("eventcalib_amd.capi", fromlist=["sync_env"]).sync_env()   # (the switches are read once per context)
            else:
                os.environ["ECAL_HOST_ARROW_PARTS"] = str(parts)
                __import__("eventcalib_amd.capi", fromlist=["sync_env"]).sync_env()   # (the switches are read once per context)
            try:
                d = np.zeros(nt)
                fail = ctypes.c_int(-1)
                rc = L.ecal_debug_arrow_solve(s._h, acc.ctypes.data, scale.ctypes.data, radius, 1e-6, 1e32, d.ctypes.data, ctypes.byref(fail), 2)
                # mode 3: the partition of the streamed evaluation (interiors shrinking towards the end of the spline)
                d3 = np.zeros(nt)
                fail3 = ctypes.c_int(-1)
                rc3 = L.ecal_debug_arrow_solve(s._h, acc.ctypes.data, scale.ctypes.data, radius, 1e-6, 1e32, d3.ctypes.data, ctypes.byref(fail3), 3)
                assert rc3 == 0 and fail3.value == 0, (rc3, fail3.value, parts)
                assert np.abs(d3 - host).max() <= 1e-9 * np.abs(host).max(), (n_cp, parts, radius, np.abs(d3 - host).max(), np.abs(host).max())
            finally:
                os.environ.pop("ECAL_HOST_ARROW_PARTS", None)
                __import__("eventcalib_amd.capi", fromlist=["sync_env"]).sync_env()   # (the switches are read once per context)
            assert rc == 0 and fail.value == 0, (rc, fail.value, parts)
            assert np.abs(d - host).max() <= 1e-9 * np.abs(host).max(), (n_cp, parts, radius, np.abs(d - host).max(), np.abs(host).max())
        # and it solves the system: (S A S + D) y = -S g, checked densely for the small cases
        if n_cp <= 50:
            perm = np.concatenate([np.arange(9, nt), np.arange(9)])
            A = H[np.ix_(perm, perm)]
            gg = g[perm]
            As = A * scale[:, None] * scale[None, :]
            D = np.clip(np.diag(As), 1e-6, 1e32) / radius
            yy = np.linalg.solve(As + np.diag(D), -gg * scale)
            assert np.abs(host - yy * scale).max() <= 1e-7 * np.abs(yy * scale).max()
    s.close()


# ---- the fisheye camera of BASELINE configs[4] (new functionality: the reference's solver refuses anything but the radial
# model, EventCalibSpline.cpp:97-99; parity = the oracle's dual numbers through the restated Kannala-Brandt functor) ----
@pytest.mark.parametrize("use_so3,n_res,n_cp", [(False, 3000, 9), (True, 2000, 6), (False, 40000, 4)])
def test_fisheye_normal_equations_match_oracle(ctx, use_so3, n_res, n_cp):
    from eventcalib_amd.capi import Solver
    rng = np.random.default_rng(n_res + 7)
    prob, x = SV.make_problem(n_res, n_cp=n_cp, seed=n_res + 7, pixel_noise=0.5, use_so3=use_so3, fisheye=True)
    y = SV.perturb(x, n_cp, rng, intr_rel=0.01, rot=0.005, trans=0.2)
    s = Solver(ctx, prob)
    cost, g, H = _dense(s.evaluate(y, True), n_cp)
    oc, og, oH = O.solver_evaluate(prob, y)
    assert abs(cost - oc) <= 1e-11 * abs(oc)
    assert np.abs(g - og).max() <= 1e-9 * np.abs(og).max()
    assert np.abs(H - oH).max() <= 1e-9 * np.abs(oH).max()
    assert abs(s.evaluate(y, False)[0] - oc) <= 1e-11 * abs(oc)
    # a different function from the radial model on the same numbers
    rc = O.solver_evaluate(dict(prob, fisheye=False), y, want_H=False)[0]
    assert abs(rc - oc) > 1e-6 * abs(oc)
    s.close()


@pytest.mark.parametrize("use_so3", [False, True])
def test_fisheye_lm_recovers_ground_truth(ctx, use_so3):
    from eventcalib_amd.capi import Solver
    rng = np.random.default_rng(15)
    n_cp = 8
    prob, x_gt = SV.make_problem(4000, n_cp=n_cp, seed=15, use_so3=use_so3, fisheye=True)
    x0 = SV.perturb(x_gt, n_cp, rng)
    s = Solver(ctx, prob)
    x, summ = s.solve(x0)
    assert summ.termination == 0 and summ.final_cost < 1e-12 * summ.initial_cost + 1e-16
    assert np.abs(x[:4] / x_gt[:4] - 1).max() < 1e-6                 # fx fy cx cy
    assert np.abs(x[4:9] - x_gt[4:9]).max() < 1e-4                   # the inverse polynomial (the high orders are weakly observed)
    xr, hist, it = ref_lm.solve(prob, x0)                            # the same minimum as the Python restatement of the LM loop
    assert np.abs(x[:9] - xr[:9]).max() <= 1e-6 * np.abs(xr[:9]).max()
    s.close()


def test_fisheye_rows_match_dual_numbers(ctx):
    import ctypes
    from eventcalib_amd.capi import Solver
    L = O.lib()
    dp = ctypes.POINTER(ctypes.c_double)
    L.oracle_find_span.argtypes = [dp, ctypes.c_uint32, ctypes.c_double]
    L.oracle_find_span.restype = ctypes.c_uint32
    L.oracle_basis.argtypes = [dp, ctypes.c_uint32, ctypes.c_double, dp]
    fn = L.oracle_residual_cam
    fn.argtypes = [dp, dp, dp, dp, dp, dp, ctypes.c_double, dp, dp, ctypes.c_int]
    fn.restype = ctypes.c_double
    n_res, n_cp = 1500, 7
    rng = np.random.default_rng(92)
    prob, x = SV.make_problem(n_res, n_cp=n_cp, seed=92, pixel_noise=0.5, fisheye=True)
    y = SV.perturb(x, n_cp, rng, intr_rel=0.01, rot=0.005, trans=0.2)
    s = Solver(ctx, prob)
    r, J, cp0 = s.residuals(y)
    intr = np.ascontiguousarray(y[:9])
    q = np.ascontiguousarray(y[9:9 + 4 * n_cp]).reshape(n_cp, 4)
    t = np.ascontiguousarray(y[9 + 4 * n_cp:]).reshape(n_cp, 3)
    kn = np.ascontiguousarray(prob["knots"], np.float64)
    lms = np.ascontiguousarray(prob["landmarks"], np.float64).reshape(-1, 3)
    p = lambda a: a.ctypes.data_as(dp)
    worst_r = worst_j = 0.0
    for k in range(0, n_res, 2):
        u = float(prob["time"][k])
        span = L.oracle_find_span(p(kn), n_cp, u)
        b4 = np.zeros(4)
        L.oracle_basis(p(kn), span, u, p(b4))
        q4 = np.ascontiguousarray(q[span - 3:span + 1]).copy()
        t4 = np.ascontiguousarray(t[span - 3:span + 1]).copy()
        obs = np.ascontiguousarray(prob["obs"][k], np.float64).copy()
        lm = lms[int(prob["lm_id"][k])].copy()
        J33 = np.zeros(33)
        rr = fn(p(intr), p(q4), p(t4), p(b4), p(obs), p(lm), float(prob["circle_radius"]), None, p(J33), 1)
        worst_r = max(worst_r, abs(r[k] - rr))
        worst_j = max(worst_j, np.abs(J[k] - J33).max() / max(1.0, np.abs(J33).max()))
    assert worst_r < 1e-11 and worst_j < 1e-10, (worst_r, worst_j)
    s.close()


# ---- the streamed evaluation of ecal_solver_solve (the interiors of the host's partition are unpacked and factorised while the
# kernel is still accumulating the later control points): same iterates as the plain form, with and without noise ----
@pytest.mark.parametrize("n_cp,n_res,parts,noise", [(60, 30000, 3, 0.0), (131, 60000, 8, 0.3), (700, 200000, None, 0.3),
                                                     (40, 20000, 5, 0.0), (300, 40, 16, 0.0)])
def test_streamed_evaluation_gives_the_plain_solves_iterates(ctx, n_cp, n_res, parts, noise, monkeypatch, capfd):
    """parts: ECAL_HOST_ARROW_PARTS (small problems take the sequential routine otherwise); (300, 40, 16): far fewer residuals than
    knot spans — groups of the stream without a single chunk, records nobody writes (an underdetermined problem held by the LM
    diagonal alone: compared to 1e-6).  The atomics' summation order differs from launch to launch, so
    'same' is 1e-9 relative on the parameters and 1e-10 on the final cost, as for the other solver variants."""
    from eventcalib_amd.capi import Solver, sync_env
    rng = np.random.default_rng(n_cp)
    prob, x_gt = SV.make_problem(n_res, n_cp=n_cp, seed=n_cp, pixel_noise=noise)
    x0 = SV.perturb(x_gt, n_cp, rng)
    if parts is not None:
        monkeypatch.setenv("ECAL_HOST_ARROW_PARTS", str(parts))
    monkeypatch.setenv("ECAL_TRACE", "solver")
    out = []
    for plain in (False, True, False):
        if plain:
            monkeypatch.setenv("ECAL_FORCE", "solver_no_stream")
        else:
            monkeypatch.delenv("ECAL_FORCE", raising=False)
        sync_env()
        s = Solver(ctx, prob)
        opt = s.default_options()
        opt.max_num_iterations = 12
        capfd.readouterr()
        x, summ = s.solve(x0, opt)
        x2, summ2 = s.solve(x0, opt)          # (again on the same solver: the group counters restore themselves)
        err = capfd.readouterr().err
        s.close()
        assert abs(summ2.final_cost - summ.final_cost) <= (1e-7 if n_res < 1000 else 1e-10) * summ.final_cost + 1e-18
        out.append((x, summ, err))
    monkeypatch.delenv("ECAL_FORCE", raising=False)
    monkeypatch.delenv("ECAL_TRACE", raising=False)
    sync_env()
    (xs, ss, es), (xp, sp, ep), (xs2, ss2, _) = out
    tol = 1e-6 if n_res < 1000 else 1e-9
    # the streamed form ran (and the plain one did not): the trace counts the streamed evaluations
    import re
    n_stream = [int(m) for m in re.findall(r"streamed evaluations: (\d+)", es)]
    n_plain = [int(m) for m in re.findall(r"streamed evaluations: (\d+)", ep)]
    assert n_stream and n_stream[0] >= 2 and n_plain and n_plain[-1] == 0, (es, ep)
    for x, summ in ((xs, ss), (xs2, ss2)):
        if noise == 0.0:
            # noise-free: the cost goes to rounding level (1e-20), where the stop tests fire on the last bits of sums whose order
            # differs from launch to launch — one iteration more or less, the same point (seen: 11 against 12)
            assert abs(summ.iterations - sp.iterations) <= 1 and abs(summ.successful_steps - sp.successful_steps) <= 1
        else:
            assert summ.iterations == sp.iterations and summ.successful_steps == sp.successful_steps
        assert abs(summ.final_cost - sp.final_cost) <= 0.1 * tol * sp.final_cost + 1e-18, (summ.final_cost, sp.final_cost)
        assert np.abs(x - xp).max() <= tol * np.abs(xp).max()


# ---- the matrix-core Gram accumulation at ragged chunk sizes (last batch of 1 .. 255 rows, 16-row steps partly or wholly empty) ----
@pytest.mark.parametrize("n_res,n_cp,fisheye", [(500, 6, False), (3000, 9, False), (40000, 4, False), (100001, 40, False), (257, 5, False),
                                                 (1, 4, False), (17, 4, False), (3000, 9, True), (70000, 4, True)])
def test_matrix_core_normal_equations_at_ragged_sizes(ctx, n_res, n_cp, fisheye):
    """v_mfma_f64_4x4x4_4b Gram accumulation (normal_eq_kernel) == the oracle's sums, and two launches agree (the chunks' atomics
    land in a different order: 1e-12)."""
    from eventcalib_amd.capi import Solver
    rng = np.random.default_rng(n_res + 1)
    prob, x = SV.make_problem(n_res, n_cp=n_cp, seed=n_res + 1, pixel_noise=0.5, fisheye=fisheye)
    y = SV.perturb(x, n_cp, rng, intr_rel=0.01, rot=0.005, trans=0.2)
    s = Solver(ctx, prob)
    acc_ws = s.evaluate(y, True).copy()
    acc_ws2 = s.evaluate(y, True).copy()
    s.close()
    scale = np.abs(acc_ws).max()
    assert scale > 0 and np.isfinite(acc_ws).all()
    tol = 1e-12 * np.maximum(np.abs(acc_ws), 1e-6 * scale)
    assert (np.abs(acc_ws2 - acc_ws) <= tol).all()
    oc, og, oH = O.solver_evaluate(prob, y)
    cost, g, H = _dense(acc_ws, n_cp)
    assert abs(cost - oc) <= 1e-11 * abs(oc) and np.abs(H - oH).max() <= 1e-10 * np.abs(oH).max()


def _acc_fields(acc, n_cp):
    """the defined entries of an accumulation buffer by kind (arrow_layout.hpp): intrinsics block (upper), per control point
    gradient, intrinsics coupling, band blocks (d = 0: upper triangle; blocks that reach beyond the last control point dropped)"""
    rec = acc[91:].reshape(n_cp, 204)
    hi = acc[10:91].reshape(9, 9)[np.triu_indices(9)]
    blocks = rec[:, 60:].reshape(n_cp, 4, 6, 6).copy()
    iu = np.tril_indices(6, -1)
    blocks[:, 0, iu[0], iu[1]] = 0.0
    for d in range(1, 4):
        blocks[n_cp - d:, d] = 0.0
    return {"cost": acc[:1], "g_intr": acc[1:10], "H_intr": hi, "g_cp": rec[:, :6], "H_cp_intr": rec[:, 6:60], "H_band": blocks}


@pytest.mark.timeout(900)
def test_benchmark_size_solve_against_the_oracle(ctx, monkeypatch):
    """bench.py's M2 problem itself — configs[2]: ONE spline of 2000 control points (12 009 unknowns) over 50 s, 45 M residuals
    (EventCalibSpline.cpp:196-247 builds one Ceres problem of that size) — not a scaled-down stand-in:
    (1) the kernel's normal equations == the oracle's dual-number rows summed on all host threads (oracle_evaluate_arrow_mt,
        ~12 s on the box's 16 CPUs), every defined entry, 1e-9 of the largest entry of its kind (sums of ~22 500 terms per
        control point in another order; measured ~1e-12);
    (2) the LM iterates of the two solve paths — streamed evaluation (the host factorises under the kernel) and plain evaluation
        (ECAL_FORCE=solver_no_stream) — agree to 1e-9 after bench.py's eight
        iterations, with the same iteration and step counts;
    (3) the streamed path is the one that ran: 16 interiors, streamed evaluations > 0, interiors found factorised."""
    import ctypes
    import synth_solver_torch as ST
    from eventcalib_amd.capi import Solver, sync_env
    n_cp, n_res, duration = 2000, 45_000_000, 50.0
    prob, x_gt = ST.make_problem(n_res, n_cp, 5.0, 5.0 + duration, seed=777, device="cuda", round_pixels=True)
    rngp = np.random.default_rng(99)                  # bench.py's start: the intrinsics off by a percent, the spline at the truth
    x0 = x_gt.copy()
    x0[:4] *= 1 + 0.01 * rngp.uniform(-1, 1, 4)
    x0[4:9] += 0.01 * rngp.uniform(-1, 1, 5)
    L = ctx._L
    L.ecal_debug_host_usable_cpus.argtypes = [ctypes.POINTER(ctypes.c_int)]
    quota = ctypes.c_int(0)
    threads = max(1, L.ecal_debug_host_usable_cpus(ctypes.byref(quota)))
    s = Solver(ctx, prob)
    assert s.n_res == n_res and s.n_cp == n_cp and s.n_params == 9 + 7 * n_cp
    # (1) normal equations
    acc = s.evaluate(x0, True)
    ref = O.solver_evaluate_arrow(prob, x0, threads)
    got_f, ref_f = _acc_fields(acc, n_cp), _acc_fields(ref, n_cp)
    worst = {}
    for k in ref_f:
        scale = np.abs(ref_f[k]).max()
        worst[k] = float(np.abs(got_f[k] - ref_f[k]).max() / scale)
        assert worst[k] <= 1e-9, (k, worst)
    assert abs(s.evaluate(x0, False)[0] - ref[0]) <= 1e-11 * ref[0]
    del ref, acc, got_f, ref_f
    # (2), (3) the three solve paths
    L.ecal_debug_solver_last_solve.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
    opt = s.default_options()
    opt.max_num_iterations = 8
    runs = {}
    for name, var in (("streamed", None), ("plain", "ECAL_FORCE")):
        if var:
            monkeypatch.setenv(var, "solver_no_stream")
        sync_env()
        try:
            x, summ = s.solve(x0, opt)
            how = (ctypes.c_uint32 * 8)()
            ctx._check(L.ecal_debug_solver_last_solve(s._h, how))
            runs[name] = (x, summ, list(how))
        finally:
            if var:
                monkeypatch.delenv(var)
            sync_env()
    s.close()
    xs, ss, hs = runs["streamed"]
    assert hs[0] == 16 and hs[1] >= 2 and hs[2] > 0 and hs[4] == 1 and hs[5] == ss.iterations, hs     # 16 interiors, streamed evaluations, interiors found factorised
    assert runs["plain"][2][0] == 16 and runs["plain"][2][1] == 0, runs["plain"][2]
    assert ss.final_cost < ss.initial_cost and np.abs(xs[:4] / x_gt[:4] - 1).max() < 8e-3
    for name in ("plain",):
        x, summ, _ = runs[name]
        assert summ.iterations == ss.iterations and summ.successful_steps == ss.successful_steps, name
        assert abs(summ.final_cost - ss.final_cost) <= 1e-10 * ss.final_cost, name
        assert np.abs(x[:9] / xs[:9] - 1).max() <= 1e-9, name                    # intrinsics, relative
        assert np.abs(x[9:] - xs[9:]).max() <= 1e-9, name                         # unit quaternions and translations (cm), absolute
