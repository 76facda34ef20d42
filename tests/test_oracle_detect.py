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


def test_full_window_loop_equals_the_single_window_functions_on_both_kd_trees():
    """oracle_detect_windows_full_mt (the batch checker of the full-size GPU tests) lays out, per window, exactly what
    event_frame + dbscan + extract_candidates give — with the restated kd-tree and with the reference's compiled one
    (oracle/_ref), which must agree with each other in every array."""
    n, rate = 120_000, 2.0e6
    buf = SS.make_stream(n, rate=rate, seed=4)
    rec = buf.numpy()
    t0, t1 = SS.tiled_windows(5.0, 5.0 + (n - 1) / rate)
    t0 = np.concatenate([t0, [t0[3], 1.0]])             # an overlapping window and an empty one
    t1 = np.concatenate([t1, [t1[5], 1.1]])
    S = len(t0)
    bounds = [O.window_bounds(rec, a, b) for a, b in zip(t0, t1)]
    size = np.array([hi - lo for lo, hi in bounds], np.uint64)
    wb = np.concatenate([[0], np.cumsum(size)]).astype(np.uint64)
    outs = {}
    try:
        for ref in ([False, True] if O.have_ref_kdtree() else [False]):
            O.set_kd_backend(ref)
            assert ("reference" in O.kd_backend()) == ref
            outs[ref] = O.detect_windows_full(rec, t0, t1, wb, int(wb[-1]), n_threads=3)
    finally:
        O.set_kd_backend(False)
    f = outs[False]
    assert f["events"] == int(size.sum())
    if True in outs:
        for k, v in f.items():
            assert np.array_equal(v, outs[True][k]), k
    paired = 0
    for s in range(S):
        lo, hi = bounds[s]
        assert (int(f["win_lo"][s]), int(f["win_hi"][s])) == (lo, hi)
        pos, neg, ep = O.event_frame(rec, lo, hi, "reference")
        b = int(wb[s])
        assert (int(f["seg_cnt"][2 * s]), int(f["seg_cnt"][2 * s + 1])) == (len(pos), len(neg))
        assert np.array_equal(f["event_point"][b:b + hi - lo], ep)
        assert np.array_equal(f["xy"][b:b + len(pos)], pos) and np.array_equal(f["xy"][b + len(pos):b + len(pos) + len(neg)], neg)
        assert f["def_pts"][b:b + len(pos) + len(neg)].all() and not f["def_pts"][b + len(pos) + len(neg):int(wb[s + 1])].any()
        for o, pts, k in ((b, pos, 0), (b + len(pos), neg, 1)):
            if len(pts):
                rc, lab, nc = O.dbscan(pts, 4.0, 2)
                assert np.array_equal(f["labels"][o:o + len(pts)], lab) and int(f["n_clusters"][2 * s + k]) == nc
        r = O.extract_candidates(pos, neg, 4.0, 2, 5, 36, 15.511363636363637)
        assert list(f["win_info"][s]) == [r["n"], r["nk_pos"], r["nk_neg"], r["status"]] and bool(f["tie"][s]) == r["tie"]
        if len(pos) and len(neg):
            assert np.array_equal(f["kept_labels"][b:b + len(pos)], r["kept_pos"]) and f["def_kept"][b:b + len(pos)].all()
            assert np.array_equal(f["kept_labels"][b + len(pos):b + len(pos) + len(neg)], r["kept_neg"])
        if not r["status"]:
            paired += 1
            assert np.array_equal(f["rep"][b:b + r["nk_pos"]], r["rep_pos"])
            assert np.array_equal(f["rep"][b + len(pos):b + len(pos) + r["nk_neg"]], r["rep_neg"])
            assert np.array_equal(f["cand_pair"][b:b + r["n"]], r["pair"]) and np.array_equal(f["cand_xyr"][b:b + r["n"]], r["xyr"])
            assert int(f["def_rep"][b:int(wb[s + 1])].sum()) == r["nk_pos"] + r["nk_neg"] and int(f["def_cand"][b:int(wb[s + 1])].sum()) == r["n"]
        else:
            assert not f["def_rep"][b:int(wb[s + 1])].any() and not f["def_cand"][b:int(wb[s + 1])].any()
    assert paired >= S // 2
