"""Vectorised (torch, CPU or GPU) generator of large synthetic solver problems — bench infrastructure.
Same model as tests/synth_solver.py: the ground truth is a uniform clamped cubic B-spline pair plus the
inverse-radial intrinsics; every generated event has residual 0 at the ground truth before the optional
rounding of the pixel to the sensor grid."""
import math

import numpy as np
import torch

import synth_solver as SV


def uniform_knots(n_cp, t0, t1):
    inner = np.linspace(t0, t1, n_cp - 2)[1:-1]
    return np.concatenate([[t0] * 4, inner, [t1] * 4])


def basis_uniform(knots_t, span, u):
    """Cox-de Boor for degree 3 (BsplineReal.hpp:107-145), vectorised: knots_t [K], span [n] long, u [n]."""
    n = u.shape[0]
    ndu = [[None] * 4 for _ in range(4)]
    ndu[0][0] = torch.ones(n, dtype=u.dtype, device=u.device)
    left = [None] * 4
    right = [None] * 4
    for j in range(1, 4):
        left[j] = u - knots_t[span + 1 - j]
        right[j] = knots_t[span + j] - u
        saved = torch.zeros_like(u)
        for r in range(j):
            ndu[j][r] = right[r + 1] + left[j - r]
            temp = ndu[r][j - 1] / ndu[j][r]
            ndu[r][j] = saved + right[r + 1] * temp
            saved = left[j - r] * temp
        ndu[j][j] = saved
    return torch.stack([ndu[0][3], ndu[1][3], ndu[2][3], ndu[3][3]], dim=1)


def quat_rotate(q, v):
    u, w = q[:, :3], q[:, 3:4]
    uv = 2 * torch.cross(u, v, dim=1)
    return v + w * uv + torch.cross(u, uv, dim=1)


def make_problem(n_res, n_cp, t0, t1, seed=0, device="cpu", round_pixels=True, intr=SV.GT_INTR, chunk=1 << 22, k_range=None):
    """Returns (problem dict of numpy arrays, ground-truth parameter vector).  k_range = (lo, hi): only the residuals
    lo .. hi - 1 of the n_res (they are in time order) — a time range of the SAME problem, for sharding one spline over
    ranks: the chunks are generated whole (seeded by index) and cut."""
    dev = torch.device(device)
    knots = uniform_knots(n_cp, t0, t1)
    q_cp, t_cp = SV.gt_control_points(n_cp, t0, t1)
    kt = torch.tensor(knots, dtype=torch.float64, device=dev)
    qc = torch.tensor(q_cp, dtype=torch.float64, device=dev)
    tc = torch.tensor(t_cp, dtype=torch.float64, device=dev)
    lms = torch.tensor(SV.landmarks(), dtype=torch.float64, device=dev)
    it = [float(v) for v in intr]
    obs_l, time_l, lm_l = [], [], []
    g = torch.Generator(device=dev)
    dt = (t1 - t0) / (n_cp - 3)
    k_lo, k_hi = (0, n_res) if k_range is None else (max(0, int(k_range[0])), min(n_res, int(k_range[1])))
    for k0 in range((k_lo // chunk) * chunk, k_hi, chunk):
        n = min(chunk, n_res - k0)
        g.manual_seed(seed * 7919 + k0 // chunk)
        # times: the k-th of n_res equally spaced instants, jittered inside its cell (sorted by construction)
        u = t0 + (torch.arange(k0, k0 + n, dtype=torch.float64, device=dev) +
                  torch.rand(n, generator=g, dtype=torch.float64, device=dev)) * ((t1 - t0) / n_res)
        u = u.clamp(max=t1)
        span = torch.clamp(((u - t0) / dt).floor().long() + 3, 3, n_cp - 1)
        # guard the half-open convention [knot_span, knot_span+1)
        span = torch.where(u < kt[span], span - 1, span)
        span = torch.where((u >= kt[span + 1]) & (span < n_cp - 1), span + 1, span)
        b = basis_uniform(kt, span, u)
        idx = span[:, None] - 3 + torch.arange(4, device=dev)[None, :]
        qv = (b[:, :, None] * qc[idx]).sum(1)
        qn = qv / qv.norm(dim=1, keepdim=True)
        T = (b[:, :, None] * tc[idx]).sum(1)
        lm = torch.randint(0, lms.shape[0], (n,), generator=g, device=dev)
        ang = torch.rand(n, generator=g, dtype=torch.float64, device=dev) * (2 * math.pi)
        Xw = lms[lm] + SV.RADIUS * torch.stack([torch.cos(ang), torch.sin(ang), torch.zeros_like(ang)], 1)
        qconj = qn * torch.tensor([-1.0, -1.0, -1.0, 1.0], dtype=torch.float64, device=dev)
        Xc = quat_rotate(qconj, Xw - T)
        pu = Xc[:, :2] / Xc[:, 2:3]
        ru = pu.norm(dim=1)
        rd = ru.clone()
        for _ in range(30):
            r2 = rd * rd
            c = 1 + it[4] * r2 + it[5] * r2 ** 2 + it[6] * r2 ** 3 + it[7] * r2 ** 4 + it[8] * r2 ** 5
            dc = 2 * rd * (it[4] + 2 * it[5] * r2 + 3 * it[6] * r2 ** 2 + 4 * it[7] * r2 ** 3 + 5 * it[8] * r2 ** 4)
            rd = rd - (rd * c - ru) / (c + rd * dc)
        pd = pu * (rd / ru.clamp_min(1e-300))[:, None]
        px = torch.stack([it[0] * pd[:, 0] + it[2], it[1] * pd[:, 1] + it[3]], 1)
        if round_pixels:
            px = torch.floor(px)          # the sensor reports integer pixels
        a, b_ = max(k_lo, k0) - k0, min(k_hi, k0 + n) - k0
        obs_l.append(px[a:b_].cpu())
        time_l.append(u[a:b_].cpu())
        lm_l.append(lm[a:b_].to(torch.int32).cpu())
    problem = dict(seg_cp_off=np.array([0, n_cp], np.uint32), knots=knots, obs=torch.cat(obs_l).numpy(),
                   time=torch.cat(time_l).numpy(), lm_id=torch.cat(lm_l).numpy().astype(np.uint32), seg_id=None,
                   landmarks=SV.landmarks(), circle_radius=SV.RADIUS, huber_a=0.2 * SV.RADIUS)
    x_gt = np.concatenate([np.asarray(intr, np.float64), q_cp.ravel(), t_cp.ravel()])
    return problem, x_gt
