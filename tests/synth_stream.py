"""Deterministic synthetic event stream (BASELINE.md §3 / SURVEY §8d) — test + bench infrastructure.

A 346x260 event camera (pinhole fx=fy=359.67525, cx=172.5, cy=129.5, radial k1..k3 =
-0.34991902, -0.014698517, 0.59684463: the constants of the reference's
event_camera_calib/test/unit_test_inverseDistortion.cpp:10-16) watches the asymmetric 9x4 circle
grid of parameter/event_calibration/example.yaml:22-37 (square 5.5 cm, radius 1.75 cm, landmarks
((2j+i%2)s, i s, 0), EventCalibIni.cpp:102-106) while moving smoothly in 6 DoF.  Event k happens at
t_k = t_start + k / rate.  90 % are edge events: a uniformly chosen circle, a uniformly chosen
angle on its rim, projected (with distortion), floored to an integer pixel; polarity 1 on the
leading half of the rim w.r.t. the circle's image velocity, 0 on the trailing half.  10 % are
uniform noise pixels with random polarity.  Records are the packed 25-byte little-endian layout of
event/include/opengv2/event/Event.hpp:41-47 (f64 t, f64 x, f64 y, u8 polarity).

Implemented with torch so the same code generates 100 k events on the CPU (tests) and 50 M events on
the GPU (bench) in chunks; the generator is seeded per chunk so results do not depend on chunking
*within one device type* (CPU and GPU random streams differ — bench data is synthetic either way).
"""
import math

import torch

SENSOR_W, SENSOR_H = 346, 260
ROWS, COLS = 9, 4
SQUARE, RADIUS = 5.5, 1.75
FX = FY = 359.67525
CX, CY = 172.5, 129.5
K1, K2, K3 = -0.34991902, -0.014698517, 0.59684463
RECORD = 25
CHUNK = 1 << 22
# BASELINE configs[4] / SURVEY 8(d): the same sensor with a fisheye lens — Kannala-Brandt, forward coefficients k1..k4
CAMERA = "pinhole"     # "pinhole": radial model above; "fisheye": theta_d = theta (1 + k1 theta^2 + .. + k4 theta^8)
KB = (0.05, -0.01, 0.002, 0.0)


def landmarks(device="cpu"):
    pts = [((2 * j + i % 2) * SQUARE, i * SQUARE, 0.0) for i in range(ROWS) for j in range(COLS)]
    return torch.tensor(pts, dtype=torch.float64, device=device)


TRAJECTORY = "hover"   # "hover": the benchmark stream (near fronto-parallel); "orbit": tilted views (see pose_orbit)


def _rot(axis, a):
    c, s, z, o = torch.cos(a), torch.sin(a), torch.zeros_like(a), torch.ones_like(a)
    rows = {0: [o, z, z, z, c, -s, z, s, c], 1: [c, z, s, z, o, z, -s, z, c], 2: [c, -s, z, s, c, z, z, z, o]}[axis]
    return torch.stack(rows, dim=1).reshape(-1, 3, 3)


def pose_orbit(t):
    """Calibration-style motion: the camera orbits the board centre with its optical axis on it, tilting up to
    ~25 degrees about both board axes (a fronto-parallel sequence leaves the focal length unobservable for
    calibrateCamera), distance 72 +- 5 cm, |omega| ~ 1 rad/s, |v| ~ 80 cm/s (inside the reference's keyframe and
    checkPose gates, EventCalibIni.cpp:82,342-343)."""
    ax = 0.40 * torch.sin(2 * math.pi * 0.43 * t + 0.3) + 0.06 * torch.sin(2 * math.pi * 1.3 * t)
    ay = 0.35 * torch.sin(2 * math.pi * 0.31 * t + 1.7) + 0.06 * torch.sin(2 * math.pi * 1.1 * t + 0.5)
    az = 0.5 * math.pi + 0.10 * torch.sin(2 * math.pi * 0.23 * t + 2.1)
    R = _rot(0, ax) @ _rot(1, ay) @ _rot(2, az)
    dist = 72.0 + 5.0 * torch.sin(2 * math.pi * 0.19 * t + 0.7)
    B = torch.tensor([3.5 * SQUARE, 4.0 * SQUARE, 0.0], dtype=t.dtype, device=t.device)
    C = B[None, :] - dist[:, None] * R[:, :, 2]
    return R, C


def pose(t):
    """Camera-to-world rotation R_wc [n,3,3] and camera centre C [n,3] (cm) at times t [n] (s)."""
    if TRAJECTORY == "orbit":
        return pose_orbit(t)
    # centre: hovering ~65 cm in front of the board (board plane z = 0, camera on the -z side)
    bx, by = 3.5 * SQUARE, 4.0 * SQUARE
    C = torch.stack([
        bx + 2.5 * torch.sin(2 * math.pi * 0.31 * t) + 1.0 * torch.sin(2 * math.pi * 0.83 * t + 0.4),
        by + 2.0 * torch.sin(2 * math.pi * 0.27 * t + 1.1) + 1.2 * torch.sin(2 * math.pi * 0.71 * t),
        -66.0 + 5.0 * torch.sin(2 * math.pi * 0.19 * t + 0.7) + 1.5 * torch.sin(2 * math.pi * 0.53 * t),
    ], dim=1)
    # small wobble (rotation vector) on top of a 90 degree roll that maps board y to image x
    w = torch.stack([
        0.06 * torch.sin(2 * math.pi * 0.37 * t + 0.3) + 0.02 * torch.sin(2 * math.pi * 1.1 * t),
        0.05 * torch.sin(2 * math.pi * 0.29 * t + 1.7) + 0.02 * torch.sin(2 * math.pi * 0.9 * t + 0.5),
        0.08 * torch.sin(2 * math.pi * 0.23 * t + 2.1),
    ], dim=1)
    th = torch.linalg.norm(w, dim=1).clamp_min(1e-12)
    k = w / th[:, None]
    K = torch.zeros(t.shape[0], 3, 3, dtype=t.dtype, device=t.device)
    K[:, 0, 1], K[:, 0, 2] = -k[:, 2], k[:, 1]
    K[:, 1, 0], K[:, 1, 2] = k[:, 2], -k[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -k[:, 1], k[:, 0]
    eye = torch.eye(3, dtype=t.dtype, device=t.device)[None]
    Rw = eye + torch.sin(th)[:, None, None] * K + (1 - torch.cos(th))[:, None, None] * (K @ K)
    roll = torch.tensor([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]], dtype=t.dtype, device=t.device)
    return Rw @ roll, C


def project(Xw, R_wc, C):
    """World points [n,3] -> distorted pixel coordinates [n,2] (forward radial model, or Kannala-Brandt: CAMERA)."""
    Xc = torch.einsum("nji,nj->ni", R_wc, Xw - C)          # R_wc^T (Xw - C)
    xn = Xc[:, 0] / Xc[:, 2]
    yn = Xc[:, 1] / Xc[:, 2]
    r2 = xn * xn + yn * yn
    if CAMERA == "fisheye":                                  # cv::fisheye::projectPoints, alpha = 0
        r = torch.sqrt(r2).clamp_min(1e-12)
        th = torch.atan(r)
        th2 = th * th
        d = th * (1 + th2 * (KB[0] + th2 * (KB[1] + th2 * (KB[2] + th2 * KB[3])))) / r
        return torch.stack([FX * xn * d + CX, FY * yn * d + CY], dim=1)
    d = 1 + K1 * r2 + K2 * r2 * r2 + K3 * r2 * r2 * r2
    return torch.stack([FX * xn * d + CX, FY * yn * d + CY], dim=1)


def _chunk(k0, n, rate, t_start, seed, device, noise_frac):
    g = torch.Generator(device=device)
    g.manual_seed(seed * 1000003 + k0 // CHUNK)
    t = t_start + (torch.arange(k0, k0 + n, dtype=torch.float64, device=device)) / rate
    u = torch.rand(n, 4, generator=g, dtype=torch.float64, device=device)
    lm = landmarks(device)
    cid = (u[:, 0] * (ROWS * COLS)).long().clamp_(max=ROWS * COLS - 1)
    ang = u[:, 1] * (2 * math.pi)
    R, C = pose(t)
    centre = lm[cid]
    rim = centre + RADIUS * torch.stack([torch.cos(ang), torch.sin(ang), torch.zeros_like(ang)], dim=1)
    px = project(rim, R, C)
    pc = project(centre, R, C)
    R2, C2 = pose(t + 1e-4)
    vel = project(centre, R2, C2) - pc
    lead = ((px - pc) * vel).sum(dim=1) > 0
    pol = lead.to(torch.uint8)
    xy = torch.floor(px)
    is_noise = u[:, 2] < noise_frac
    out = (xy[:, 0] < 0) | (xy[:, 0] >= SENSOR_W) | (xy[:, 1] < 0) | (xy[:, 1] >= SENSOR_H)
    is_noise = is_noise | out
    nz = torch.rand(n, 3, generator=g, dtype=torch.float64, device=device)
    nxy = torch.stack([torch.floor(nz[:, 0] * SENSOR_W), torch.floor(nz[:, 1] * SENSOR_H)], dim=1)
    xy = torch.where(is_noise[:, None], nxy, xy)
    pol = torch.where(is_noise, (nz[:, 2] < 0.5).to(torch.uint8), pol)
    return t, xy, pol


def pack_records(t, xy, pol):
    n = t.shape[0]
    rec = torch.empty(n, RECORD, dtype=torch.uint8, device=t.device)
    rec[:, 0:8] = t.contiguous().view(torch.uint8).reshape(n, 8)
    rec[:, 8:24] = xy.contiguous().view(torch.uint8).reshape(n, 16)
    rec[:, 24] = pol
    return rec.reshape(-1)


def make_stream(n_events, rate=1.0e6, t_start=5.0, seed=12345, device="cpu", noise_frac=0.1, k_offset=0, total=None):
    """Packed .bin image of the stream: uint8 tensor of n_events*25 bytes on `device`.  k_offset / total: the events
    k_offset .. k_offset + n_events - 1 of the `total`-event stream that starts at t_start — a time range of it, for sharding
    ONE stream over ranks: the chunks are generated as the whole stream generates them (seeded by their index, sized by
    `total`) and cut, so the slice holds the same records whoever generates it."""
    total = k_offset + n_events if total is None else total
    parts = []
    k_end = k_offset + n_events
    for k0 in range((k_offset // CHUNK) * CHUNK, k_end, CHUNK):
        t, xy, pol = _chunk(k0, min(CHUNK, total - k0), rate, t_start, seed, device, noise_frac)
        lo, hi = max(k0, k_offset) - k0, min(k0 + CHUNK, k_end) - k0
        parts.append(pack_records(t[lo:hi], xy[lo:hi], pol[lo:hi]))
    return torch.cat(parts) if len(parts) > 1 else parts[0]


def unpack_records(buf):
    """uint8 [n*25] (torch, any device) -> (t f64 [n], xy f64 [n,2], pol u8 [n])."""
    n = buf.numel() // RECORD
    r = buf.reshape(n, RECORD)
    t = r[:, 0:8].contiguous().view(torch.float64).reshape(n)
    xy = r[:, 8:24].contiguous().view(torch.float64).reshape(n, 2)
    return t, xy, r[:, 24].contiguous()


def tiled_windows(t_first, t_last, length=1.5e-3):
    """Back-to-back inclusive windows [t0, t1] covering [t_first, t_last] (policy P1).
    t1 is the largest double below the next window's t0, so every event is in exactly one window."""
    import numpy as np
    n = int(math.floor((t_last - t_first) / length)) + 1
    t0 = t_first + length * np.arange(n, dtype=np.float64)
    t1 = np.nextafter(t_first + length * np.arange(1, n + 1, dtype=np.float64), -np.inf)
    return t0, t1


def undistortion_error_px(intr, radius=0.30):
    """Max error, in pixels of the generating camera, of a refined camera's pixel -> undistorted-ray map
    (EventCalibSpline.hpp:194-204: x = (u - cx) / fx, X = x (1 + k1 r^2 + .. + k5 r^10)) against the generating radial model,
    over the rays within `radius` (tan of the angle to the optical axis; 0.30 = 108 px around the principal point: where the
    board's circles are seen — beyond it the five-term polynomial extrapolates, as the reference's would).  The events are floored
    to integer pixels, so a generated point observed at pixel p is p - 0.5 to the estimator (the principal point comes back 0.5
    low)."""
    import numpy as np
    xn, yn = np.meshgrid(np.linspace(-radius, radius, 121), np.linspace(-radius, radius, 121))
    r2 = xn * xn + yn * yn
    d = 1 + K1 * r2 + K2 * r2 * r2 + K3 * r2 * r2 * r2
    u, v = FX * xn * d + CX - 0.5, FY * yn * d + CY - 0.5
    keep = (u >= 0) & (u <= SENSOR_W - 1) & (v >= 0) & (v <= SENSOR_H - 1) & (r2 <= radius * radius)
    fx, fy, cx, cy = intr[:4]
    x, y = (u - cx) / fx, (v - cy) / fy
    q2 = x * x + y * y
    c = 1 + q2 * (intr[4] + q2 * (intr[5] + q2 * (intr[6] + q2 * (intr[7] + q2 * intr[8]))))
    err = FX * np.hypot(x * c - xn, y * c - yn)
    return float(err[keep].max())
