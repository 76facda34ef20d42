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
