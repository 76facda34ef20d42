"""Dense numpy restatement of the Levenberg-Marquardt loop the product implements (Ceres 1.x
trust-region minimiser with Jacobi scaling) on top of the solver oracle — test infrastructure."""
import numpy as np

import oracle_lib as O


def quat_plus(q, d):
    nd = np.linalg.norm(d)
    if nd > 0:
        dq = np.concatenate([np.sin(nd) / nd * d, [np.cos(nd)]])
    else:
        dq = np.array([0.0, 0.0, 0.0, 1.0])
    a, b = dq, q
    return np.array([a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1], a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0],
                     a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3], a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2]])


def so3_plus(q, d):
    """LocalParameterizationSO3::Plus: q (x) exp(d), exp with the half angle (Sophus)."""
    nd = np.linalg.norm(d)
    e = np.concatenate([np.sin(nd / 2) / nd * d, [np.cos(nd / 2)]]) if nd > 0 else np.array([0.0, 0.0, 0.0, 1.0])
    a, b = q, e
    return np.array([a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1], a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0],
                     a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3], a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2]])


def plus(x, d, n_cp, use_so3=False):
    y = x.copy()
    y[:9] += d[:9]
    rot_plus = so3_plus if use_so3 else quat_plus
    for c in range(n_cp):
        y[9 + 4 * c: 13 + 4 * c] = rot_plus(x[9 + 4 * c: 13 + 4 * c], d[9 + 6 * c: 12 + 6 * c])
        y[9 + 4 * n_cp + 3 * c: 12 + 4 * n_cp + 3 * c] += d[12 + 6 * c: 15 + 6 * c]
    return y


def solve(problem, x0, max_iter=50, ftol=1e-10, gtol=1e-10, ptol=1e-8):
    n_cp = int(problem["seg_cp_off"][-1])
    x = x0.copy()
    cost, g, H = O.solver_evaluate(problem, x)
    scale = 1.0 / (1.0 + np.sqrt(np.diag(H)))
    radius, dec = 1e4, 2.0
    hist = [cost]
    it = 0
    while it < max_iter:
        it += 1
        Hs = H * scale[:, None] * scale[None, :]
        dd = np.clip(np.diag(Hs), 1e-6, 1e32) / radius
        try:
            Lc = np.linalg.cholesky(Hs + np.diag(dd))
        except np.linalg.LinAlgError:
            radius /= dec
            dec *= 2
            continue
        ys = -np.linalg.solve(Lc.T, np.linalg.solve(Lc, g * scale))
        d = ys * scale
        model = -(g @ d) - 0.5 * d @ H @ d
        if model <= 0:
            radius /= dec
            dec *= 2
            continue
        xn = plus(x, d, n_cp, bool(problem.get("use_so3", False)))
        new_cost = O.solver_evaluate(problem, xn, want_H=False)[0]
        rel = (cost - new_cost) / model
        if rel > 1e-3:
            change, prev = cost - new_cost, cost
            x = xn
            cost, g, H = O.solver_evaluate(problem, x)
            hist.append(cost)
            t = 2 * rel - 1
            radius = min(1e16, radius / max(1.0 / 3.0, 1 - t ** 3))
            dec = 2.0
            if np.abs(g).max() <= gtol or abs(change) <= ftol * prev:
                break
        else:
            radius /= dec
            dec *= 2
        if np.linalg.norm(d) <= ptol * (np.linalg.norm(x) + ptol):
            break
    return x, hist, it
