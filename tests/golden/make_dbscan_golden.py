"""Generates tests/golden/dbscan_*.npz (run in the build container, where /root/reference exists).

Expected labels come from the oracle's Run()/expandCluster() restatement running on top of the
REFERENCE's own kd-tree (oracle/_ref/libkdtree_ref.so, compiled from
/root/reference/modules/camera_calibration/dbscan/src/kdtree.cpp by oracle/Makefile) and are
asserted equal to the fully restated oracle before being written.  Fixtures are data only:
points, eps, minpts, expected labels and cluster counts.

    python tests/golden/make_dbscan_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402
import synth  # noqa: E402


def run_case(name, xy, off, eps, minpts):
    xy = np.ascontiguousarray(xy, dtype=np.float64)
    off = np.asarray(off, dtype=np.uint32)
    labels = np.full(xy.shape[0], -1, np.int32)
    ncl = np.zeros(len(off) - 1, np.uint32)
    for s in range(len(off) - 1):
        seg = xy[off[s]:off[s + 1]]
        if seg.shape[0] == 0:
            continue
        rc, l_ref, n_ref = O.dbscan(seg, eps, minpts, kdapi=True)
        rc2, l_own, n_own = O.dbscan(seg, eps, minpts)
        assert rc == 0 and rc2 == 0 and n_ref == n_own and (l_ref == l_own).all(), name
        labels[off[s]:off[s + 1]] = l_ref
        ncl[s] = n_ref
    path = os.path.join(HERE, "dbscan_%s.npz" % name)
    np.savez_compressed(path, xy=xy, off=off, eps=np.float64(eps), minpts=np.uint32(minpts), labels=labels,
                        n_clusters=ncl)
    print("%-28s points %6d slices %3d clusters %5d  %6.1f KB" % (
        name, xy.shape[0], len(off) - 1, int(ncl.sum()), os.path.getsize(path) / 1024))


def main():
    assert O.have_ref_kdtree(), "build oracle/_ref first (make -C oracle)"
    rng = np.random.default_rng(20201011)
    xy, off = synth.arc_slices(rng, 6, 0.1)
    run_case("arcs_eps4_quirk", xy, off, 4.0, 2)          # shipped config (example.yaml:68-71), quirk active
    xy, off = synth.arc_slices(rng, 4, 0.5)
    run_case("arcs_noisy_eps4", xy, off, 4.0, 2)
    run_case("arcs_noisy_eps4p5", xy, off, 4.5, 2)        # quirk-free
    xy, off = synth.arc_slices(rng, 2, 2.0)
    run_case("arcs_heavy_noise_minpts5", xy, off, 4.0, 5)
    for lat, eps in ((0.5, 2.5), (0.1, 0.3), (0.1, 1.7), (1.0, 3.0), (1.0, 5.0)):
        parts = [synth.random_segment(rng, int(rng.integers(1, 500)), lat, 30.0) for _ in range(8)]
        off = np.zeros(9, np.uint32)
        off[1:] = np.cumsum([p.shape[0] for p in parts])
        run_case("lattice%s_eps%s" % (str(lat).replace(".", "p"), str(eps).replace(".", "p")),
                 np.concatenate(parts), off, eps, int(rng.choice([1, 2, 5])))
    # edge cases: empty, n=1, all noise, one giant cluster, duplicates only, axis-aligned exact-eps chains
    chain = np.stack([np.arange(0, 200, 4.0), np.zeros(50)], 1)
    rng.shuffle(chain, axis=0)
    parts = [np.zeros((0, 2)), np.array([[3.0, 3.0]]), np.array([[0.0, 0.0], [100.0, 100.0], [200.0, 0.0]]),
             np.stack([np.arange(300.0), np.zeros(300)], 1), np.tile(np.array([[5.0, 5.0]]), (40, 1)), chain,
             np.array([[4.0, 5.0], [4.0, 0.0], [0.0, 0.0]]), np.zeros((0, 2))]
    off = np.zeros(len(parts) + 1, np.uint32)
    off[1:] = np.cumsum([p.shape[0] for p in parts])
    run_case("edge_cases_minpts1", np.concatenate(parts), off, 4.0, 1)
    run_case("edge_cases_minpts2", np.concatenate(parts), off, 4.0, 2)


if __name__ == "__main__":
    main()
