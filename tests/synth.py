"""Synthetic inputs for tests and bench (test infrastructure; deterministic given the rng/seed).

* arc_slices      — per-polarity pixel sets of a 4x9 asymmetric circle grid (half arcs + noise),
                    the shape CirclesEventFrame::extractFeatures hands to DBSCAN::Run
                    (event_camera_calib/src/CirclesEventFrame.cpp:66-72); integer pixels, random order.
* random_segments — adversarial point sets (lattices 1 / 0.5 / 0.1, duplicates, tiny and empty
                    segments) for the quirk / edge-case parity tests.
Sensor and pattern constants follow parameter/event_calibration/example.yaml:22-37.
"""
import numpy as np

SENSOR_W, SENSOR_H = 346, 260
ROWS, COLS = 9, 4
SQUARE, RADIUS = 5.5, 1.75


def grid_centres(square):
    """Pattern landmarks ((2j + i%2)*s, i*s) (EventCalibIni.cpp:102-106), as an [36,2] array."""
    pts = []
    for i in range(ROWS):
        for j in range(COLS):
            pts.append(((2 * j + i % 2) * square, i * square))
    return np.asarray(pts, dtype=np.float64)


def one_arc_slice(rng, noise_frac=0.1, which=0):
    """Unique integer pixels of one polarity of one slice: 36 half arcs + uniform noise, shuffled."""
    s = rng.uniform(26.0, 31.0)            # px per pattern square (pattern long side along image width)
    r = RADIUS / SQUARE * s
    c = grid_centres(s)
    c = np.stack([c[:, 1], c[:, 0]], axis=1)   # rows along x
    ang = rng.uniform(-0.15, 0.15)
    R = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
    c = (c - c.mean(0)) @ R.T + np.array([SENSOR_W / 2, SENSOR_H / 2]) + rng.uniform(-8, 8, size=2)
    mdir = rng.uniform(0, 2 * np.pi) + (np.pi if which else 0.0)
    th = mdir + np.linspace(-np.pi / 2, np.pi / 2, int(np.pi * r * 1.6))
    pts = (c[:, None, :] + r * np.stack([np.cos(th), np.sin(th)], axis=1)[None, :, :]).reshape(-1, 2)
    pts = pts + rng.normal(0, 0.35, size=pts.shape)
    n_noise = int(noise_frac * pts.shape[0])
    noise = np.stack([rng.uniform(0, SENSOR_W, n_noise), rng.uniform(0, SENSOR_H, n_noise)], axis=1)
    px = np.floor(np.concatenate([pts, noise], axis=0))
    ok = (px[:, 0] >= 0) & (px[:, 0] < SENSOR_W) & (px[:, 1] >= 0) & (px[:, 1] < SENSOR_H)
    px = np.unique(px[ok], axis=0)
    rng.shuffle(px, axis=0)
    return px.astype(np.float64)


def arc_slices(rng, n_slices, noise_frac=0.1):
    """n_slices polarity-slices back to back.  Returns (xy [N,2] f64, slice_off [n_slices+1] u32)."""
    parts = [one_arc_slice(rng, noise_frac, k & 1) for k in range(n_slices)]
    off = np.zeros(n_slices + 1, dtype=np.uint32)
    off[1:] = np.cumsum([p.shape[0] for p in parts])
    xy = np.concatenate(parts, axis=0) if parts else np.zeros((0, 2))
    return np.ascontiguousarray(xy), off


def random_segment(rng, n, lattice, span):
    if lattice == 0.1:
        return rng.integers(0, int(span * 10) + 1, size=(n, 2)).astype(np.float64) * 0.1
    return np.round(rng.uniform(0, span, size=(n, 2)) / lattice) * lattice


def random_segments(rng, n_segments, max_n=300, lattices=(1.0, 0.5, 0.1), spans=(10, 30, 80), with_empty=True):
    """Mixed bag of segments (one lattice per batch so a single eps applies).  Returns (xy, off)."""
    lattice = float(rng.choice(lattices))
    parts = []
    for _ in range(n_segments):
        r = rng.random()
        if with_empty and r < 0.05:
            n = 0
        elif r < 0.15:
            n = int(rng.integers(1, 4))
        else:
            n = int(rng.integers(1, max_n + 1))
        p = random_segment(rng, n, lattice, float(rng.choice(spans)))
        if n > 4 and rng.random() < 0.3:   # inject exact duplicates
            k = int(rng.integers(1, n // 2 + 1))
            p[rng.integers(0, n, k)] = p[rng.integers(0, n, k)]
        parts.append(p)
    off = np.zeros(n_segments + 1, dtype=np.uint32)
    off[1:] = np.cumsum([p.shape[0] for p in parts])
    xy = np.concatenate(parts, axis=0) if parts else np.zeros((0, 2))
    return np.ascontiguousarray(xy), off, lattice
