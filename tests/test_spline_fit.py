"""CPU: ecal_spline_fit / ecal_spline_eval (host code of libecal.so, no GPU involved) against the numpy restatement
of BsplineReal's approximating constructor and against scipy's independent B-spline basis.  Tolerance 1e-10
relative: banded Cholesky vs dense solve of the same normal equations."""
import os
import sys

import numpy as np
import pytest
from scipy.interpolate import BSpline

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import spline_fit_oracle as SO  # noqa: E402


def _samples(m, seed, dim):
    rng = np.random.default_rng(seed)
    u = np.sort(rng.uniform(5.0, 7.0, m))
    u[0] -= 1.5e-3
    u[-1] += 1.5e-3                       # EventCalibSpline.cpp:63-66
    t = (u - u[0]) / (u[-1] - u[0])
    Q = np.stack([np.sin(3 * t + k) + 0.3 * np.cos(11 * t * (k + 1)) for k in range(dim)], 1) * 20
    return u, Q + 0.05 * rng.normal(size=Q.shape)


@pytest.mark.parametrize("m,n_cp,dim", [(400, 80, 3), (400, 80, 4), (12, 4, 3), (12, 5, 3), (50, 25, 4), (9, 6, 3)])
def test_fit_matches_restatement(m, n_cp, dim):
    from eventcalib_amd import capi
    u, Q = _samples(m, m + n_cp, dim)
    kn, cp = capi.spline_fit(u, Q, n_cp)
    kn_o, cp_o = SO.fit(u, Q, n_cp)
    assert np.array_equal(kn, kn_o)
    assert np.allclose(cp, cp_o, rtol=1e-10, atol=1e-10 * np.abs(cp_o).max())
    assert np.array_equal(cp[0], Q[0]) and np.array_equal(cp[-1], Q[-1])      # endpoints interpolate
    # evaluation agrees with scipy's B-spline on the same knots / control points
    uu = np.linspace(u[0], u[-1], 257)
    ours = capi.spline_eval(kn, cp, uu)
    ref = BSpline(kn, cp, 3)(uu)
    assert np.allclose(ours, ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())


def test_fit_reproduces_a_spline_in_its_own_space():
    """Samples of a cubic spline on the fit's own knot vector come back exactly (least squares with zero residual)."""
    from eventcalib_amd import capi
    rng = np.random.default_rng(3)
    u = np.sort(rng.uniform(0, 1, 300))
    n_cp = 40
    kn = SO.knot_vector(u, n_cp)
    cp_true = rng.normal(size=(n_cp, 3))
    Q = BSpline(kn, cp_true, 3)(np.clip(u, kn[0], kn[-1]))
    cp_true[0], cp_true[-1] = Q[0], Q[-1]
    Q = BSpline(kn, cp_true, 3)(u)
    kn2, cp = capi.spline_fit(u, Q, n_cp)
    assert np.allclose(cp, cp_true, atol=1e-8)


def test_fit_error_paths():
    from eventcalib_amd import capi
    u, Q = _samples(20, 1, 3)
    with pytest.raises(capi.EcalError):
        capi.spline_fit(u, Q, 3)                       # n_cp <= degree (:79-82)
    with pytest.raises(capi.EcalError):
        capi.spline_fit(u[::-1].copy(), Q, 5)          # parameters not ascending
    kn, cp = capi.spline_fit(u, Q, 6)
    with pytest.raises(capi.EcalError):
        capi.spline_eval(kn, cp, [u[-1] + 1.0])        # outside the bound


# ---- BsplineSO3::optimizeCP (ecal_spline_so3_refine) --------------------------------------------------------------
def _qmul(a, b):
    return np.array([a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1], a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0],
                     a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3], a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2]])


def _qexp(w):
    th = np.linalg.norm(w)
    if th < 1e-10:
        return np.array([0.5 * w[0], 0.5 * w[1], 0.5 * w[2], 1.0 - th * th / 8])
    return np.concatenate([np.sin(th / 2) / th * w, [np.cos(th / 2)]])


def _qlog(q):
    n = np.linalg.norm(q[:3])
    if n < 1e-10:
        return 2.0 / q[3] * q[:3]
    return 2.0 * np.arctan(n / q[3]) / n * q[:3]


def _qinv(q):
    return np.array([-q[0], -q[1], -q[2], q[3]])


def _so3_residuals(kn, cp, S, u):
    """P3ApproximationError (BsplineSO3.hpp:121-153) restated: r_i = log(S_i^-1 cp0 exp(b1 log(cp0^-1 cp1)) ...)."""
    n_cp = len(cp)
    out = np.zeros((len(u), 3))
    for i, ui in enumerate(u):
        span = min(max(int(np.searchsorted(kn, ui, side="right")) - 1, 3), n_cp - 1)
        N = np.array([BSpline.basis_element(kn[span - 3 + j: span + 2 + j], extrapolate=False)(ui) if kn[span - 3 + j] < kn[span + 1 + j]
                      else 0.0 for j in range(4)])
        N = np.nan_to_num(N)
        if ui >= kn[-1]:                                   # right end of the clamped knot vector: N_{n-1} = 1
            N = np.array([0.0, 0.0, 0.0, 1.0])
        beta = [N[1] + N[2] + N[3], N[2] + N[3], N[3]]
        X = cp[span - 3].copy()
        for j in range(1, 4):
            X = _qmul(X, _qexp(beta[j - 1] * _qlog(_qmul(_qinv(cp[span - 4 + j]), cp[span - 3 + j]))))
        out[i] = _qlog(_qmul(_qinv(S[i]), X))
    return out


def _so3_problem(n_cp, m, seed, noise):
    rng = np.random.default_rng(seed)
    u = np.sort(rng.uniform(5.0, 6.0, m))
    # a smooth rotation trajectory sampled at u (+ noise on the group)
    S = np.stack([_qexp(np.array([0.8 * np.sin(2.1 * (t - 5)), 0.5 * np.cos(3.3 * (t - 5)), 0.6 * (t - 5)])) for t in u])
    if noise > 0:
        S = np.stack([_qmul(s, _qexp(noise * rng.normal(size=3))) for s in S])
    return u, S


@pytest.mark.parametrize("n_cp,m,noise", [(8, 60, 0.0), (12, 150, 0.01), (16, 160, 0.02), (4, 20, 0.01)])
def test_so3_refine_reaches_the_minimum_of_the_restated_problem(n_cp, m, noise):
    """ecal_spline_so3_refine against scipy's least_squares on the numpy restatement of the same residuals, unknowns = rotation
    vector steps cp <- cp exp(delta) of the interior control points: same minimum (cost to 1e-8 relative, control points to
    1e-6), stationary (finite-difference gradient of the restated cost at the returned point)."""
    from eventcalib_amd import capi
    from scipy.optimize import least_squares
    u, S = _so3_problem(n_cp, m, 7 * n_cp + m, noise)
    kn, cp0 = capi.spline_fit(u, S, n_cp)                      # BsplineSO3::initialGuess: the fit on the quaternion coefficients
    cp0 /= np.linalg.norm(cp0, axis=1, keepdims=True)
    cp, info = capi.spline_so3_refine(kn, cp0, S, u)
    cost0 = 0.5 * (_so3_residuals(kn, cp0, S, u) ** 2).sum()
    cost1 = 0.5 * (_so3_residuals(kn, cp, S, u) ** 2).sum()
    assert np.isclose(info["initial_cost"], cost0, rtol=1e-9, atol=1e-14) and np.isclose(info["final_cost"], cost1, rtol=1e-9, atol=1e-14)
    assert cost1 <= cost0 * (1 + 1e-12) and np.allclose(np.linalg.norm(cp, axis=1), 1.0, atol=1e-12)
    assert np.array_equal(cp[0], cp0[0]) and np.array_equal(cp[-1], cp0[-1])          # SetParameterBlockConstant (:294-295)

    def with_steps(base, d):
        out = base.copy()
        for c in range(1, n_cp - 1):
            out[c] = _qmul(base[c], _qexp(d[3 * (c - 1): 3 * c]))
        return out
    nu = 3 * (n_cp - 2)
    if nu == 0:
        return
    # stationarity at the returned point
    g = np.zeros(nu)
    for a in range(nu):
        e = np.zeros(nu)
        e[a] = 1e-6
        g[a] = (0.5 * (_so3_residuals(kn, with_steps(cp, e), S, u) ** 2).sum() - 0.5 * (_so3_residuals(kn, with_steps(cp, -e), S, u) ** 2).sum()) / 2e-6
    assert np.abs(g).max() < 1e-6 * max(1.0, cost0)
    # an independent minimiser from the same start
    ref = least_squares(lambda d: _so3_residuals(kn, with_steps(cp0, d), S, u).ravel(), np.zeros(nu), xtol=1e-14, ftol=1e-14, gtol=1e-12)
    cp_ref = with_steps(cp0, ref.x)
    assert abs(cost1 - ref.cost) <= 1e-8 * max(ref.cost, 1e-12) + 1e-13
    sign = np.sign((cp * cp_ref).sum(1))
    assert np.abs(cp - cp_ref * sign[:, None]).max() < 1e-6


def test_so3_refine_recovers_a_spline_in_its_own_space():
    from eventcalib_amd import capi
    rng = np.random.default_rng(11)
    n_cp, m = 10, 200
    u = np.sort(rng.uniform(0.0, 1.0, m))
    kn = SO.knot_vector(u, n_cp)
    cp_true = np.stack([_qexp(0.4 * rng.normal(size=3)) for _ in range(n_cp)])
    # samples = the cumulative spline itself: residuals vanish at cp_true
    S = np.stack([_qmul(_qexp(np.zeros(3)), q) for q in cp_true])       # placeholder shape
    from_spline = np.zeros((m, 4))
    ident = np.array([0.0, 0.0, 0.0, 1.0])
    r = _so3_residuals(kn, cp_true, np.tile(ident, (m, 1)), u)          # log(X(u_i)) with S = identity
    for i in range(m):
        from_spline[i] = _qexp(r[i])
    start = cp_true.copy()
    for c in range(1, n_cp - 1):
        start[c] = _qmul(cp_true[c], _qexp(0.05 * rng.normal(size=3)))
    cp, info = capi.spline_so3_refine(kn, start, from_spline, u)
    assert info["final_cost"] < 1e-20 and info["initial_cost"] > 1e-4
    sign = np.sign((cp * cp_true).sum(1))
    assert np.abs(cp - cp_true * sign[:, None]).max() < 1e-8


def test_so3_refine_error_paths():
    from eventcalib_amd import capi
    u, S = _so3_problem(8, 40, 1, 0.0)
    kn, cp0 = capi.spline_fit(u, S, 8)
    with pytest.raises(capi.EcalError):
        capi.spline_so3_refine(kn, cp0, S, u + 10.0)               # samples outside the knot range
    bad = cp0.copy()
    bad[3] = 0
    with pytest.raises(capi.EcalError):
        capi.spline_so3_refine(kn, bad, S, u)                      # a zero quaternion
