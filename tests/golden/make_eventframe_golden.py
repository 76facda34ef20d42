"""Generates tests/golden/eventframe_order_*.npz (run in the build container, where /root/reference exists and the
toolchain is g++ 11.4 / libstdc++ GLIBCXX_3.4.29 — the point order below is that library's).

Per fixture: the packed 25-byte records of a few windows -> per window the reference's positiveEvents_ /
negativeEvents_ (EventFrame.cpp:10-36 run on a real std::unordered_set with the restated EigenMatrixHash,
oracle_event_frame_ref) -> DBSCAN labels of both lists from the oracle's Run() on the REFERENCE's kd-tree
(oracle/_ref/libkdtree_ref.so = kdtree.cpp compiled where it lies).  Data only: records, window index ranges, expected
points, event -> point map, labels, cluster counts.

    python tests/golden/make_eventframe_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402
import synth_stream as SS  # noqa: E402


def run_case(name, rec, bounds, eps, minpts):
    rec = np.ascontiguousarray(rec, np.uint8)
    xy_all, ep_all, lab_all, cnt, ncl = [], [], [], [], []
    for lo, hi in bounds:
        pos, neg, ep = O.event_frame(rec, lo, hi, "reference")
        mp, mn, mep = O.event_frame(rec, lo, hi, "model")
        assert np.array_equal(pos, mp) and np.array_equal(neg, mn) and np.array_equal(ep, mep), name
        ep_all.append(ep)
        for pts in (pos, neg):
            xy_all.append(pts)
            cnt.append(pts.shape[0])
            if pts.shape[0]:
                rc, lab, nc = O.dbscan(pts, eps, minpts, kdapi=True)
                rc2, lab2, nc2 = O.dbscan(pts, eps, minpts)
                assert rc == 0 and rc2 == 0 and nc == nc2 and np.array_equal(lab, lab2), name
            else:
                lab, nc = np.zeros(0, np.int32), 0
            lab_all.append(lab)
            ncl.append(nc)
    path = os.path.join(HERE, "eventframe_order_%s.npz" % name)
    np.savez_compressed(path, records=rec, bounds=np.asarray(bounds, np.int64), eps=np.float64(eps), minpts=np.uint32(minpts),
                        xy=np.concatenate(xy_all).reshape(-1, 2), seg_cnt=np.asarray(cnt, np.uint32),
                        event_point=np.concatenate(ep_all), labels=np.concatenate(lab_all), n_clusters=np.asarray(ncl, np.uint32))
    print("%-26s events %6d windows %3d points %6d clusters %5d  %6.1f KB" % (
        name, rec.size // 25, len(bounds), sum(cnt), sum(ncl), os.path.getsize(path) / 1024))


def main():
    assert O.have_ref_kdtree(), "build oracle/_ref first (make -C oracle)"
    # the benchmark stream's first windows (1 Mev/s, 1.5 ms tiles): ~1500 events, ~650 keys per polarity -> 7 epochs
    buf = SS.make_stream(9000, device="cpu").numpy()
    t = buf.reshape(-1, 25)[:, :8].copy().view(np.float64).reshape(-1)
    t0, t1 = SS.tiled_windows(float(t[0]), float(t[-1]))
    bounds = [O.window_bounds(buf, a, b) for a, b in zip(t0, t1)]
    run_case("stream_1mevs", buf, bounds, 4.0, 2)
    # denser stream: windows of ~3000 events (second pass of the hash slicer; 1109 < keys -> the 2357-bucket epoch)
    buf = SS.make_stream(9000, rate=2.0e6, device="cpu", seed=21).numpy()
    t = buf.reshape(-1, 25)[:, :8].copy().view(np.float64).reshape(-1)
    t0, t1 = SS.tiled_windows(float(t[0]), float(t[-1]))
    bounds = [O.window_bounds(buf, a, b) for a, b in zip(t0, t1)]
    run_case("stream_2mevs", buf, bounds, 4.0, 2)
    # every epoch boundary: windows whose positive set holds exactly B-1, B, B+1 keys for B = 13 ... 541 (+ cancellations)
    rng = np.random.default_rng(20201011)
    recs, bounds, k0 = [], [], 0
    for B in (13, 29, 59, 127, 257, 541):
        for m in (B - 1, B, B + 1):
            pix = rng.choice(346 * 260, m + 40, replace=False)
            x, y = (pix % 346).astype(float), (pix // 346).astype(float)
            pol = np.ones(m + 40, np.uint8)
            pol[m:] = 0                                    # 40 negative keys ...
            order = rng.permutation(m + 40)
            x, y, pol = x[order], y[order], pol[order]
            dup = rng.integers(0, m + 40, 60)              # ... 60 repeats (no-ops) ...
            cx, cy, cp = x[dup], y[dup], pol[dup]
            flip = rng.random(60) < 0.3                    # ... some with the other polarity: that pixel is erased
            cp = np.where(flip, 1 - cp, cp).astype(np.uint8)
            X, Y, P = np.concatenate([x, cx]), np.concatenate([y, cy]), np.concatenate([pol, cp])
            n = X.shape[0]
            recs.append(O.pack_events(k0 * 1e-6 + np.arange(n) * 1e-6, X, Y, P))
            bounds.append((k0, k0 + n))
            k0 += n
    run_case("epoch_boundaries", np.concatenate(recs), bounds, 4.0, 2)
    # general tiers: half-pixel and negative coordinates, -0.0
    n = 2600
    x = rng.integers(-20, 60, n).astype(float) * rng.choice([1.0, 0.5], n)
    y = rng.integers(-10, 40, n).astype(float)
    x[rng.random(n) < 0.02] = -0.0
    p = (rng.random(n) < 0.5).astype(np.uint8)
    run_case("general_coordinates", O.pack_events(np.arange(n) * 1e-6, x, y, p), [(0, 900), (900, 2600), (0, 2600)], 2.5, 2)


if __name__ == "__main__":
    main()
