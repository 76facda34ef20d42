"""CPU suite: pin the oracle (own restatement vs the reference's compiled kdtree.cpp vs the closed
form the HIP kernel implements vs scikit-learn) and the golden fixtures."""
import glob
import os

import numpy as np
import pytest

import closed_form
import oracle_lib as O
import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
needs_ref = pytest.mark.skipif(not O.have_ref_kdtree(), reason="oracle/_ref/libkdtree_ref.so not built")


def test_survey_three_point_demo():
    # A=(4,5), B=(4,0), C=(0,0), eps 4, minpts 1 — insertion order changes the result
    A, B, C = (4.0, 5.0), (4.0, 0.0), (0.0, 0.0)
    rc, lab, nc = O.dbscan(np.array([A, B, C]), 4.0, 1)
    assert rc == 0 and nc == 1 and lab.tolist() == [-1, 0, -1]
    rc, lab, nc = O.dbscan(np.array([B, C, A]), 4.0, 1)
    assert nc == 1 and lab.tolist() == [0, 0, -1]
    rc, lab, nc = O.dbscan(np.array([C, B, A]), 4.0, 1)
    assert nc == 1 and lab.tolist() == [0, 0, -1]


def test_failed_inputs():
    rc, lab, nc = O.dbscan(np.zeros((0, 2)), 4.0, 2)
    assert rc == 1 and nc == 0          # dbscan.h:121
    rc, lab, nc = O.dbscan(np.zeros((3, 2)), 4.0, 0)
    assert rc == 1                      # dbscan.h:123


@needs_ref
@pytest.mark.parametrize("seed", range(6))
def test_range_query_matches_reference_kdtree(seed):
    """Own tree + walk == reference kd_insert/kd_nearest_range, hit for hit, in list order."""
    rng = np.random.default_rng(100 + seed)
    for _ in range(25):
        xy, off, lattice = synth.random_segments(rng, 1, max_n=500, with_empty=False)
        eps = float(rng.choice([3, 4, 4.5, 5, 2.5, 1.7, 0.3]))
        o1, i1 = O.range_query_all(xy, eps)
        o2, i2 = O.range_query_all(xy, eps, kdapi=True)
        assert (o1 == o2).all() and (i1 == i2).all()


@needs_ref
@pytest.mark.parametrize("seed", range(4))
def test_driver_on_reference_kdtree(seed):
    """Restated Run()/expandCluster() on top of the reference kd_* ABI == fully restated path,
    including the element order inside every cluster."""
    rng = np.random.default_rng(200 + seed)
    for _ in range(20):
        xy, off, lattice = synth.random_segments(rng, 1, max_n=600, with_empty=False)
        eps = float(rng.choice([3, 4, 4.5, 5, 2.5, 1.7]))
        minpts = int(rng.choice([1, 2, 5]))
        rc1, l1, n1, c1 = O.dbscan(xy, eps, minpts, with_members=True)
        rc2, l2, n2, c2 = O.dbscan(xy, eps, minpts, kdapi=True, with_members=True)
        assert rc1 == rc2 == 0 and n1 == n2 and (l1 == l2).all()
        assert all((a == b).all() for a, b in zip(c1, c2))


@needs_ref
def test_arc_slices_on_reference_kdtree():
    rng = np.random.default_rng(5)
    for noise in (0.1, 0.5):
        xy = synth.one_arc_slice(rng, noise)
        rc1, l1, n1 = O.dbscan(xy, 4.0, 2)
        rc2, l2, n2 = O.dbscan(xy, 4.0, 2, kdapi=True)
        assert n1 == n2 and (l1 == l2).all() and n1 >= 30


@pytest.mark.parametrize("seed", range(6))
def test_closed_form_equals_oracle(seed):
    """The data-parallel formulation (kd-cell bounds + directed reachability) == sequential oracle."""
    rng = np.random.default_rng(300 + seed)
    fired = 0
    for _ in range(30):
        xy, off, lattice = synth.random_segments(rng, 1, max_n=350, with_empty=False)
        eps = float(rng.choice([3, 4, 4.5, 5, 2.5, 1.7, 0.3]))
        minpts = int(rng.choice([1, 2, 5]))
        rc, lab, nc = O.dbscan(xy, eps, minpts)
        lab2, nc2 = closed_form.dbscan(xy, eps, minpts)
        assert nc == nc2 and (lab == lab2).all()
        # how often does the strict-pruning quirk matter?  (exact-ball model differs)
        E = closed_form.edges(xy, eps)
        d = xy[None, :, :] - xy[:, None, :]
        ball = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) <= eps * eps
        np.fill_diagonal(ball, False)
        fired += int((ball != E).any())
    assert fired > 0, "test inputs never exercised the pruning quirk"


def test_closed_form_on_arcs():
    rng = np.random.default_rng(11)
    xy = synth.one_arc_slice(rng, 0.3)
    rc, lab, nc = O.dbscan(xy, 4.0, 2)
    lab2, nc2 = closed_form.dbscan(xy, 4.0, 2)
    assert nc == nc2 and (lab == lab2).all()


def test_sklearn_cross_check_quirk_free():
    """Third opinion where the quirk cannot fire (no lattice pair at distance exactly eps):
    sklearn's core set and core partition must match (border points are Noise in the reference)."""
    sk = pytest.importorskip("sklearn.cluster")
    rng = np.random.default_rng(17)
    for _ in range(10):
        xy = synth.one_arc_slice(rng, float(rng.choice([0.1, 0.5])))
        eps, minpts = 4.5, 2
        rc, lab, nc = O.dbscan(xy, eps, minpts)
        m = sk.DBSCAN(eps=eps, min_samples=minpts + 1).fit(xy)   # sklearn counts the point itself
        core = np.zeros(xy.shape[0], bool)
        core[m.core_sample_indices_] = True
        assert ((lab >= 0) == core).all()
        # same partition of core points
        a = lab[core]
        b = m.labels_[core]
        pairs = set(zip(a.tolist(), b.tolist()))
        assert len(pairs) == len(set(a.tolist())) == len(set(b.tolist()))


def test_golden_fixtures():
    files = sorted(glob.glob(os.path.join(GOLDEN, "dbscan_*.npz")))
    assert files, "no golden fixtures committed"
    for f in files:
        z = np.load(f)
        lab, ncl = O.dbscan_batch(z["xy"], z["off"][:-1], np.diff(z["off"]).astype(np.uint32), float(z["eps"]),
                                  int(z["minpts"]))
        assert (lab == z["labels"]).all(), f
        assert (ncl == z["n_clusters"]).all(), f
