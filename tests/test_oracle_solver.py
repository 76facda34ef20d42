"""CPU suite: the solver oracle against known answers, and the product's analytic residual/Jacobian
header (compiled for the host with g++) against the oracle's dual-number differentiation."""
import os
import subprocess

import numpy as np

import oracle_lib as O
import synth_solver as SV

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_inverse_radial_kat():
    # constants of unit_test_inverseDistortion.cpp:10-16 through PinholeCamera.cpp:74-92
    b = O.inverse_radial([-0.34991902, -0.014698517, 0.59684463, 0.0])
    want = [0.34991902, 0.38202847867328127, -0.041555343865844696, -1.1638270394205459, -4.138165444396021]
    assert np.allclose(b, want, rtol=1e-14, atol=0)


def test_ground_truth_has_zero_cost_and_gradient():
    prob, x = SV.make_problem(300, n_cp=8, seed=1)
    cost, g, H = O.solver_evaluate(prob, x)
    assert cost < 1e-18 and np.abs(g).max() < 1e-7
    assert np.allclose(H, H.T) and np.linalg.eigvalsh(H).min() > -1e-6


def test_gradient_matches_finite_differences():
    rng = np.random.default_rng(2)
    prob, x = SV.make_problem(200, n_cp=7, seed=2, pixel_noise=0.3)
    y = SV.perturb(x, 7, rng, intr_rel=0.005, rot=0.002, trans=0.05)
    cost, g, _ = O.solver_evaluate(prob, y, want_H=False)
    # central differences on the intrinsics and the translations (Euclidean blocks)
    for idx in list(range(9)) + [9 + 4 * 7 + k for k in (0, 4, 11, 20)]:
        h = 1e-6 * max(1.0, abs(y[idx]))
        yp, ym = y.copy(), y.copy()
        yp[idx] += h
        ym[idx] -= h
        fd = (O.solver_evaluate(prob, yp, False)[0] - O.solver_evaluate(prob, ym, False)[0]) / (2 * h)
        gi = g[idx] if idx < 9 else g[9 + 6 * ((idx - 9 - 28) // 3) + 3 + (idx - 9 - 28) % 3]
        assert abs(fd - gi) <= 1e-5 * (1 + abs(fd)), (idx, fd, gi)


def test_product_residual_header_matches_oracle(tmp_path):
    exe = str(tmp_path / "check_residual")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "check_residual.cpp"),
                           "-L" + O.ORACLE_DIR, "-loracle", "-Wl,-rpath," + O.ORACLE_DIR])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def _dense_from_acc(acc, n_cp):
    """the accumulation-buffer layout (arrow_layout.hpp) back to cost, gradient and the dense symmetric matrix in the oracle's
    tangent order [intr 9 | (rot 3, trans 3) per control point]"""
    N = 9 + 6 * n_cp
    g = np.zeros(N)
    H = np.zeros((N, N))
    g[:9] = acc[1:10]
    for i in range(9):
        for j in range(i, 9):
            H[i, j] = H[j, i] = acc[10 + 9 * i + j]
    for c in range(n_cp):
        rec = acc[91 + 204 * c: 91 + 204 * (c + 1)]
        g[9 + 6 * c: 15 + 6 * c] = rec[:6]
        H[9 + 6 * c: 15 + 6 * c, :9] = rec[6:60].reshape(6, 9)
        H[:9, 9 + 6 * c: 15 + 6 * c] = rec[6:60].reshape(6, 9).T
        for d in range(4):
            if c + d >= n_cp:
                break
            blk = rec[60 + 36 * d: 96 + 36 * d].reshape(6, 6)
            if d == 0:
                blk = np.triu(blk) + np.triu(blk, 1).T
            H[9 + 6 * c: 15 + 6 * c, 9 + 6 * (c + d): 15 + 6 * (c + d)] = blk
            H[9 + 6 * (c + d): 15 + 6 * (c + d), 9 + 6 * c: 15 + 6 * c] = blk.T
    return acc[0], g, H


def test_arrow_layout_evaluation_equals_the_dense_one():
    """oracle_evaluate_arrow_mt (the checker of the GPU normal equations at benchmark size, any thread count) == the dense
    oracle_evaluate_mode it restates in the accumulation buffer's layout: radial and fisheye camera, both rotation variants"""
    rng = np.random.default_rng(5)
    for kw in ({}, {"fisheye": True}, {"use_so3": True}):
        prob, x = SV.make_problem(3000, n_cp=11, seed=7, pixel_noise=0.4, **kw)
        y = SV.perturb(x, 11, rng, intr_rel=0.01, rot=0.004, trans=0.1)
        cost, g, H = O.solver_evaluate(prob, y)
        for threads in (1, 3, 8):
            c2, g2, H2 = _dense_from_acc(O.solver_evaluate_arrow(prob, y, threads), 11)
            assert abs(c2 - cost) <= 1e-12 * cost
            assert np.abs(g2 - g).max() <= 1e-11 * np.abs(g).max()
            assert np.abs(H2 - H).max() <= 1e-11 * np.abs(H).max()
