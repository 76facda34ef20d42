"""CPU suite: the host's linear solves of the LM step (eventcalib_amd/csrc/ecal_solver.hip solve_arrow, arrow_host_parts.hpp) on
random banded-arrow systems in the accumulation-buffer layout — the sequential routine, the partitioned one (blocked banded
Cholesky per interior, separators eliminated one after the other, intrinsics' corner) on both partitions — against a dense numpy
solve of (S A S + D) y = -S g.  No GPU involved: ecal_debug_arrow_solve_host runs on the host alone."""
import ctypes

import numpy as np
import pytest

import eventcalib_amd

ACC_HEAD, ACC_PER_CP = 91, 204


def _system(n_cp, seed, rows_per_span=5):
    """J^T J of a Jacobian with the solver's sparsity: every row touches the 9 intrinsics and 4 consecutive control points."""
    rng = np.random.default_rng(seed)
    nt = 9 + 6 * n_cp
    H = np.zeros((nt, nt))
    g = np.zeros(nt)
    for span in range(3, n_cp):
        cols = np.concatenate([np.arange(9), 9 + 6 * (span - 3) + np.arange(24)])
        J = rng.standard_normal((rows_per_span, 33)) * np.concatenate([np.full(9, 0.3), np.tile([1.0, 2.0, 0.5, 30.0, 20.0, 10.0], 4)])
        r = rng.standard_normal(rows_per_span)
        H[np.ix_(cols, cols)] += J.T @ J
        g[cols] += J.T @ r
    acc = np.zeros(ACC_HEAD + ACC_PER_CP * n_cp)
    acc[0] = 1.0
    acc[1:10] = g[:9]
    for i in range(9):
        for j in range(i, 9):
            acc[10 + 9 * i + j] = H[i, j]
    for c in range(n_cp):
        b = ACC_HEAD + ACC_PER_CP * c
        acc[b:b + 6] = g[9 + 6 * c:15 + 6 * c]
        acc[b + 6:b + 60] = H[9 + 6 * c:15 + 6 * c, :9].reshape(-1)
        for d in range(4):
            if c + d >= n_cp:
                break
            blk = H[9 + 6 * c:15 + 6 * c, 9 + 6 * (c + d):15 + 6 * (c + d)].copy()
            if d == 0:
                blk = np.triu(blk)
            acc[b + 60 + 36 * d:b + 96 + 36 * d] = blk.reshape(-1)
    return H, g, acc


def _hook():
    L = eventcalib_amd.load_library()
    L.ecal_debug_arrow_solve_host.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                              ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int]
    L.ecal_debug_arrow_solve_host.restype = ctypes.c_int
    return L


@pytest.mark.parametrize("n_cp,parts_list", [(28, [2, 4]), (45, [3, 6]), (131, [2, 5, 8, 16]), (400, [7, 16, 32])])
def test_host_arrow_solves_agree_with_a_dense_solve(n_cp, parts_list):
    L = _hook()
    H, g, acc = _system(n_cp, seed=n_cp)
    nt = 9 + 6 * n_cp
    perm = np.concatenate([np.arange(9, nt), np.arange(9)])      # the solver's order: control points, then intrinsics
    A = H[np.ix_(perm, perm)]
    gg = g[perm]
    scale = 1.0 / (1.0 + np.sqrt(np.diag(A)))
    for radius in (1e4, 3.0, 1e-2):
        As = A * scale[:, None] * scale[None, :]
        D = np.clip(np.diag(As), 1e-6, 1e32) / radius
        want = np.linalg.solve(As + np.diag(D), -gg * scale) * scale

        def run(mode, parts=0):
            d = np.zeros(nt)
            fail = ctypes.c_int(-1)
            rc = L.ecal_debug_arrow_solve_host(n_cp, acc.ctypes.data, scale.ctypes.data, radius, 1e-6, 1e32, d.ctypes.data, ctypes.byref(fail), mode, parts)
            assert rc == 0 and fail.value == 0, (rc, fail.value, mode, parts)
            return d
        seq = run(0)
        tol = 1e-9 * np.abs(want).max()
        assert np.abs(seq - want).max() <= tol, (n_cp, radius, np.abs(seq - want).max(), np.abs(want).max())
        for parts in parts_list:
            for mode in (2, 3):
                d = run(mode, parts)
                assert np.abs(d - want).max() <= tol, (n_cp, radius, mode, parts, np.abs(d - want).max(), np.abs(want).max())
                assert np.abs(d - seq).max() <= 1e-10 * np.abs(seq).max()


def test_an_indefinite_system_is_reported_not_solved():
    L = _hook()
    n_cp = 60
    H, g, acc = _system(n_cp, seed=3)
    # a negative diagonal entry deep inside an interior and one inside a separator
    for c in (17, n_cp // 2):
        bad = acc.copy()
        bad[ACC_HEAD + ACC_PER_CP * c + 60] = -1e9
        nt = 9 + 6 * n_cp
        scale = np.ones(nt)
        for mode, parts in ((0, 0), (2, 4), (3, 4), (2, 8), (3, 8)):
            d = np.zeros(nt)
            fail = ctypes.c_int(-1)
            rc = L.ecal_debug_arrow_solve_host(n_cp, bad.ctypes.data, scale.ctypes.data, 1e4, 1e-6, 1e32, d.ctypes.data, ctypes.byref(fail), mode, parts)
            assert rc == 0 and fail.value == 1, (rc, fail.value, mode, parts, c)


def test_partitions_cover_the_control_points():
    """both partitions: interiors + 3-control-point separators = the spline; the streamed one ends in small interiors"""
    L = eventcalib_amd.load_library()
    L.ecal_debug_arrow_partition.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    L.ecal_debug_arrow_partition.restype = ctypes.c_int
    for n_cp, P in ((2000, 16), (700, 7), (131, 16), (60, 3), (28, 4), (5000, 32)):
        for stream in (0, 1):
            f = np.zeros(P, np.uint32)
            m = np.zeros(P, np.uint32)
            assert L.ecal_debug_arrow_partition(n_cp, P, stream, f.ctypes.data, m.ctypes.data) == 0
            assert f[0] == 0 and (m >= 4).all(), (n_cp, P, stream, m)
            assert (f[1:] == f[:-1] + m[:-1] + 3).all() and f[-1] + m[-1] == n_cp
            if stream:
                assert (np.diff(m.astype(np.int64)) <= 1).all()          # never growing towards the end
                if n_cp >= 700:
                    assert m[-1] <= max(6, 0.007 * n_cp) + 1 and m[0] > 4 * m[-1]


@pytest.mark.parametrize("workers", [0, 1, 3, 15])
def test_the_worker_pool_runs_every_task_once(workers):
    """HostPool (arrow_host_parts.hpp): runs of 1 - 40 tasks handed out among parked threads that poll, sleep and are nudged awake
    — from the caller and from inside a task, as the streamed evaluation does — every task exactly once, none of another run."""
    L = eventcalib_amd.load_library()
    L.ecal_debug_host_pool_selftest.argtypes = [ctypes.c_int, ctypes.c_int]
    L.ecal_debug_host_pool_selftest.restype = ctypes.c_int
    assert L.ecal_debug_host_pool_selftest(workers, 3000) == 0
