"""TEST INFRASTRUCTURE ONLY — numpy restatement of the reference's approximating B-spline constructor.

Follows core/spline/include/opengv2/spline/BsplineReal.hpp: knot placement :87-100 (NURBS book 9.68), findSpan
:208-231, basis functions :107-145 (A2.2), the endpoint-interpolating least squares :329-449 — written densely
(the reference uses Eigen sparse matrices + SimplicialLDLT; same normal equations).  Eigen is absent from the
image, so the header itself cannot be compiled here: parity unpinned against a reference run, pinned by
scipy.interpolate.BSpline (independent basis evaluation) and by exact reproduction of splines that lie in the
fitted space (tests/test_spline_fit.py)."""
import numpy as np

P = 3


def knot_vector(u, n_cp):
    m = len(u)
    kn = np.empty(n_cp + P + 1)
    kn[:P + 1] = u[0]
    kn[-(P + 1):] = u[-1]
    d = m / float(n_cp - P)
    for j in range(1, n_cp - P):
        i = int(np.floor(j * d))
        a = j * d - i
        kn[P + j] = (1 - a) * u[i - 1] + a * u[i]
    return kn


def find_span(kn, n_cp, u):
    n = n_cp - 1
    if u == kn[n + 1]:
        return n
    lo, hi = P, n + 1
    mid = (lo + hi) // 2
    while u < kn[mid] or u >= kn[mid + 1]:
        if u < kn[mid]:
            hi = mid
        else:
            lo = mid
        mid = (lo + hi) // 2
    return mid


def basis(kn, span, u):
    left, right = np.zeros(P + 1), np.zeros(P + 1)
    N = np.zeros(P + 1)
    N[0] = 1.0
    for j in range(1, P + 1):
        left[j] = u - kn[span + 1 - j]
        right[j] = kn[span + j] - u
        saved = 0.0
        for r in range(j):
            temp = N[r] / (right[r + 1] + left[j - r])
            N[r] = saved + right[r + 1] * temp
            saved = left[j - r] * temp
        N[j] = saved
    return N


def design(kn, n_cp, u):
    A = np.zeros((len(u), n_cp))
    for k, uk in enumerate(u):
        s = find_span(kn, n_cp, uk)
        A[k, s - P:s + 1] = basis(kn, s, uk)
    return A


def fit(u, Q, n_cp):
    u, Q = np.asarray(u, float), np.asarray(Q, float)
    kn = knot_vector(u, n_cp)
    N = design(kn, n_cp, u)
    cp = np.zeros((n_cp, Q.shape[1]))
    cp[0], cp[-1] = Q[0], Q[-1]
    R = Q[1:-1] - np.outer(N[1:-1, 0], Q[0]) - np.outer(N[1:-1, -1], Q[-1])
    Nc = N[1:-1, 1:-1]
    cp[1:-1] = np.linalg.solve(Nc.T @ Nc, Nc.T @ R)
    return kn, cp
