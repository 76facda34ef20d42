"""CPU suite: known answers for the candidate-extraction oracle (oracle/detect_oracle.cpp)."""
import numpy as np

import oracle_lib as O
import synth_stream as SS


def test_circle_radius_threshold_kat():
    # shipped config 346x260, 9x4 asymmetric, square 5.5, radius 1.75
    v = O.circle_radius_threshold(346.0, 260.0, 9, 4, True, 5.5, 1.75)
    assert v == 15.511363636363637
    # symmetric branch (CirclesEventFrame.cpp:27-31)
    v2 = O.circle_radius_threshold(346.0, 260.0, 9, 4, False, 5.5, 1.75)
    assert abs(v2 - min(346.0 / 9, 260.0 / 4) / 5.5 * 1.75 * 1.5) < 1e-15


def test_fit_circle_exact_points():
    th = np.linspace(0, 2 * np.pi, 24, endpoint=False)
    pts = np.stack([40 + 7.5 * np.cos(th), 30 + 7.5 * np.sin(th)], 1)
    c, r = O.fit_circle(pts[:12], pts[12:])
    assert np.allclose(c, [40, 30], atol=1e-9) and abs(r - 7.5) < 1e-9


def test_two_half_arcs_make_one_candidate():
    """36 clean circles: every +/- half-arc pair is mutually nearest -> 36 candidates on the grid."""
    rng = np.random.default_rng(3)
    cx, cy = np.meshgrid(30 + 42.0 * np.arange(9), 25 + 45.0 * np.arange(4))   # gap between circles > diameter
    centres = np.stack([cx.ravel(), cy.ravel()], 1)
    r = 9.0
    pos, neg = [], []
    for c in centres:
        th = np.linspace(-1.4, 1.4, 26)
        pos.append(np.unique(np.floor(c + r * np.stack([np.cos(th), np.sin(th)], 1)), axis=0))
        neg.append(np.unique(np.floor(c + r * np.stack([np.cos(th + np.pi), np.sin(th + np.pi)], 1)), axis=0))
    pos = np.concatenate(pos)
    neg = np.concatenate(neg)
    rng.shuffle(pos, axis=0)
    rng.shuffle(neg, axis=0)
    out = O.extract_candidates(pos, neg, 4.0, 2, 5, 36, 15.5)
    assert out["status"] == 0 and out["nk_pos"] == 36 and out["nk_neg"] == 36 and out["n"] == 36
    d = np.linalg.norm(out["xyr"][:, None, :2] - (centres[None] - 0.5), axis=2).min(1)
    # centre = midpoint of the two norm-median pixels: a crude estimate by design (refined by rectifyFeatures)
    assert (d < 8.0).all() and (np.abs(out["xyr"][:, 2] - r) < 4.0).all()
    # fitCircle == 1 path (:180-281): the algebraic fit recovers the true centres / radius much more closely
    fit = O.extract_candidates(pos, neg, 4.0, 2, 5, 36, 15.5, fit_circle=True, knn_num=3)
    assert fit["status"] == 0 and fit["n"] == 36
    d = np.linalg.norm(fit["xyr"][:, None, :2] - (centres[None] - 0.5), axis=2).min(1)
    assert (d < 0.8).all() and (np.abs(fit["xyr"][:, 2] - r) < 0.8).all()
    # too few clusters -> extractFeatures returns false (:127-129)
    out = O.extract_candidates(pos[:200], neg, 4.0, 2, 5, 36, 15.5)
    assert out["status"] == 1 and out["n"] == 0
    out = O.extract_candidates(np.zeros((0, 2)), neg, 4.0, 2, 5, 36, 15.5)
    assert out["status"] == 1 and out["nk_neg"] == 0     # :62-64 returns before DBSCAN


def test_synthetic_window_gives_candidates():
    buf = SS.make_stream(30000, rate=2.0e6)     # denser stream -> complete half arcs
    t, _, _ = SS.unpack_records(buf)
    rec = buf.numpy()
    lo, hi = O.window_bounds(rec, float(t[0]), float(t[0]) + 1.5e-3)
    pos, neg, _ = O.event_frame(rec, lo, hi, "reference")
    out = O.extract_candidates(pos, neg, 4.0, 2, 5, 36, 15.511363636363637)
    assert out["status"] == 0 and out["n"] >= 30
