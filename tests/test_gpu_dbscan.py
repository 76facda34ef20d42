"""GPU parity: HIP DBSCAN (through the C ABI) vs the oracle, bit-exact labels."""
import glob
import os

import numpy as np
import pytest

import oracle_lib as O
import synth

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ctx():
    import eventcalib_amd
    c = eventcalib_amd.Context(0)
    yield c
    c.close()


def _check(ctx, xy, off, eps, minpts):
    labels, ncl = ctx.dbscan_batch(xy, off, eps, minpts)
    ref_l, ref_n = O.dbscan_batch(xy, off[:-1], np.diff(off).astype(np.uint32), eps, minpts)
    bad = np.nonzero(labels != ref_l)[0]
    assert bad.size == 0, "%d/%d labels differ (first at %d: got %d want %d)" % (
        bad.size, labels.size, bad[0], labels[bad[0]], ref_l[bad[0]])
    assert (ncl == ref_n).all()


def test_golden_fixtures(ctx):
    files = sorted(glob.glob(os.path.join(GOLDEN, "dbscan_*.npz")))
    assert files
    for f in files:
        z = np.load(f)
        labels, ncl = ctx.dbscan_batch(z["xy"], z["off"], float(z["eps"]), int(z["minpts"]))
        assert (labels == z["labels"]).all(), f
        assert (ncl == z["n_clusters"]).all(), f


def test_which_kd_tree_checked_the_kernels(ctx):
    """States (and asserts) what sits under the oracle when it judges the HIP kernels in this session: with oracle/_ref
    present it is the reference's own kdtree.cpp — then the kernels are checked against BOTH trees here, which must agree with
    each other; without it the session is checked by the restated tree only, and this test says so by being skipped."""
    rng = np.random.default_rng(5)
    xy, off = synth.arc_slices(rng, 96, 0.5)
    cnt = np.diff(off).astype(np.uint32)
    labels, ncl = ctx.dbscan_batch(xy, off, 4.0, 2)
    print("\n[parity] oracle k-d tree of this session: %s" % O.kd_backend())
    if not O.have_ref_kdtree():
        ref_l, ref_n = O.dbscan_batch(xy, off[:-1], cnt, 4.0, 2)
        assert np.array_equal(labels, ref_l) and np.array_equal(ncl, ref_n)
        pytest.skip("oracle/_ref/libkdtree_ref.so absent: the kernels of this session are checked by the RESTATED k-d tree only")
    assert "reference" in O.kd_backend()
    try:
        res = {}
        for ref in (True, False):
            O.set_kd_backend(ref)
            res[ref] = O.dbscan_batch(xy, off[:-1], cnt, 4.0, 2)
            assert np.array_equal(labels, res[ref][0]) and np.array_equal(ncl, res[ref][1]), O.kd_backend()
        assert np.array_equal(res[True][0], res[False][0])
    finally:
        O.set_kd_backend(True)


@pytest.mark.parametrize("noise", [0.0, 0.1, 0.5, 2.0])
def test_arc_slices(ctx, noise):
    rng = np.random.default_rng(int(noise * 10) + 1)
    xy, off = synth.arc_slices(rng, 64, noise)
    _check(ctx, xy, off, 4.0, 2)


@pytest.mark.parametrize("seed", range(8))
def test_random_lattices_quirk(ctx, seed):
    rng = np.random.default_rng(1000 + seed)
    xy, off, lattice = synth.random_segments(rng, 200, max_n=400)
    for eps in (4.0, 3.0, 5.0, 2.5, 1.7, 4.5, 0.3):
        for minpts in (1, 2, 5):
            _check(ctx, xy, off, eps, minpts)


def test_edge_cases(ctx):
    # n = 1, all-noise, one giant cluster, duplicates only, empty slices at both ends
    parts = [np.zeros((0, 2)), np.array([[3.0, 3.0]]), np.array([[0.0, 0.0], [100.0, 100.0], [200.0, 0.0]]),
             np.stack([np.arange(300.0), np.zeros(300)], 1), np.tile(np.array([[5.0, 5.0]]), (40, 1)),
             np.zeros((0, 2))]
    off = np.zeros(len(parts) + 1, np.uint32)
    off[1:] = np.cumsum([p.shape[0] for p in parts])
    xy = np.concatenate(parts)
    for minpts in (1, 2, 5):
        _check(ctx, xy, off, 4.0, minpts)
    labels, ncl = ctx.dbscan_batch(xy, off, 4.0, 2)
    assert ncl.tolist() == [0, 0, 0, 1, 1, 0]


def test_size_tiers_and_big_path(ctx):
    """Segments that land in every tier: <=1024, <=2048, <=4096 (LDS) and > 4096 (global scratch)."""
    rng = np.random.default_rng(77)
    sizes = [1000, 1024, 1025, 2048, 2049, 4096, 4097, 9000, 30]
    parts = []
    for n in sizes:
        span = int(np.sqrt(n) * 3) + 4
        parts.append(rng.integers(0, span, size=(n, 2)).astype(np.float64))
    off = np.zeros(len(parts) + 1, np.uint32)
    off[1:] = np.cumsum(sizes)
    xy = np.concatenate(parts)
    _check(ctx, xy, off, 4.0, 2)
    _check(ctx, xy, off, 3.0, 5)


def test_degenerate_orders_and_coordinates(ctx):
    rng = np.random.default_rng(3)
    # row-major sorted input: the insertion-order tree degenerates to depth O(n)
    g = np.stack(np.meshgrid(np.arange(40.0), np.arange(20.0)), -1).reshape(-1, 2)
    keep = rng.random(g.shape[0]) < 0.5
    sorted_pts = g[keep]
    # huge coordinates: the cell hash is unusable -> brute-force candidate scan
    far = rng.integers(0, 50, size=(200, 2)).astype(np.float64) + 1.0e13
    neg = rng.integers(-60, 60, size=(300, 2)).astype(np.float64)
    parts = [sorted_pts, far, neg]
    off = np.zeros(len(parts) + 1, np.uint32)
    off[1:] = np.cumsum([p.shape[0] for p in parts])
    _check(ctx, np.concatenate(parts), off, 4.0, 2)


def test_invalid_arguments(ctx):
    import eventcalib_amd
    with pytest.raises(eventcalib_amd.EcalError):
        ctx.dbscan_batch(np.zeros((3, 2)), np.array([0, 3], np.uint32), 4.0, 0)   # Run() FAILED: min < 1
    for bad_eps in (0.0, -4.0, float("nan"), float("inf")):
        with pytest.raises(eventcalib_amd.EcalError):
            ctx.dbscan_batch(np.zeros((3, 2)), np.array([0, 3], np.uint32), bad_eps, 2)
    labels, ncl = ctx.dbscan_batch(np.zeros((0, 2)), np.array([0], np.uint32), 4.0, 2)
    assert labels.size == 0 and ncl.size == 0


def test_pixel_and_general_kernels_agree(ctx):
    """Integer pixel segments take the lean pixel kernel (dbscan_pixel.hpp); with ECAL_FORCE=dbscan_general the general
    tiers do the same segments.  Both must equal the oracle — including the pruning quirk (integral eps), duplicate
    pixels (pixel kernel bails), a bounding box too large for its bitmap, and segments above its 1024-point capacity."""
    rng = np.random.default_rng(77)
    segs = []
    for k in range(40):                                   # dense lattice patches: many exactly-eps pairs
        n = int(rng.integers(20, 900))
        segs.append(np.stack([rng.integers(0, 60, n), rng.integers(0, 45, n)], 1).astype(np.float64))
        segs[-1] = np.unique(segs[-1], axis=0)[rng.permutation(len(np.unique(segs[-1], axis=0)))]
    dup = np.stack([rng.integers(0, 30, 300), rng.integers(0, 30, 300)], 1).astype(np.float64)   # duplicates
    segs.append(dup)
    wide = np.stack([rng.integers(-8000, 8000, 500), rng.integers(-8000, 8000, 500)], 1).astype(np.float64)
    wide[:200] = np.stack([rng.integers(0, 40, 200), rng.integers(0, 40, 200)], 1)                # bbox too large
    segs.append(wide)
    big = np.unique(np.stack([rng.integers(0, 120, 3000), rng.integers(0, 90, 3000)], 1), axis=0).astype(np.float64)
    segs.append(big[rng.permutation(len(big))][:1500])                                            # > 1024 points
    row = np.stack([np.arange(300), np.full(300, 7)], 1).astype(np.float64)                        # > 255 points in a row
    segs.append(row[rng.permutation(300)])
    xy = np.concatenate(segs)
    off = np.concatenate([[0], np.cumsum([len(s) for s in segs])]).astype(np.uint32)
    for eps, minpts in ((4.0, 2), (3.0, 1), (5.0, 5), (4.5, 2), (15.0, 3), (16.0, 2)):
        os.environ.pop("ECAL_FORCE", None)
        __import__("eventcalib_amd.capi", fromlist=["sync_env"]).sync_env()   # (the switches are read once per context)
        la, na = ctx.dbscan_batch(xy, off, eps, minpts)
        os.environ["ECAL_FORCE"] = "dbscan_general"
        __import__("eventcalib_amd.capi", fromlist=["sync_env"]).sync_env()   # (the switches are read once per context)
        try:
            lb, nb = ctx.dbscan_batch(xy, off, eps, minpts)
        finally:
            os.environ.pop("ECAL_FORCE", None)
            __import__("eventcalib_amd.capi", fromlist=["sync_env"]).sync_env()   # (the switches are read once per context)
        assert np.array_equal(la, lb) and np.array_equal(na, nb), (eps, minpts)
        ref_l, ref_n = O.dbscan_batch(xy, off[:-1], np.diff(off).astype(np.uint32), eps, minpts)
        assert np.array_equal(la, ref_l) and np.array_equal(na, ref_n), (eps, minpts)


def test_compiled_and_generic_disc_agree(ctx):
    """floor(eps^2) == 16 (the shipped eps = 4) runs the pixel kernel with the disc compiled in (packed half-disc word,
    multiply-free anchor test); ECAL_FORCE=dbscan_generic_disc forces the run-time disc on the same input.  Sparse noise
    around dense arcs gives core points with non-core neighbours; eps = 4.1 has the same disc without the quirk."""
    rng = np.random.default_rng(5)
    segs = []
    for k in range(60):
        n = int(rng.integers(50, 1000))
        dense = np.stack([rng.integers(0, 50, n), rng.integers(0, 40, n)], 1)
        sparse = np.stack([rng.integers(0, 340, n // 3 + 1), rng.integers(0, 250, n // 3 + 1)], 1)
        p = np.unique(np.concatenate([dense, sparse]), axis=0).astype(np.float64)
        segs.append(p[rng.permutation(len(p))][:1024])
    xy = np.concatenate(segs)
    off = np.concatenate([[0], np.cumsum([len(s) for s in segs])]).astype(np.uint32)
    for eps, minpts in ((4.0, 2), (4.0, 1), (4.0, 5), (4.1, 2)):
        os.environ.pop("ECAL_FORCE", None)
        __import__("eventcalib_amd.capi", fromlist=["sync_env"]).sync_env()   # (the switches are read once per context)
        la, na = ctx.dbscan_batch(xy, off, eps, minpts)
        os.environ["ECAL_FORCE"] = "dbscan_generic_disc"
        __import__("eventcalib_amd.capi", fromlist=["sync_env"]).sync_env()   # (the switches are read once per context)
        try:
            lb, nb = ctx.dbscan_batch(xy, off, eps, minpts)
        finally:
            os.environ.pop("ECAL_FORCE", None)
            __import__("eventcalib_amd.capi", fromlist=["sync_env"]).sync_env()   # (the switches are read once per context)
        assert np.array_equal(la, lb) and np.array_equal(na, nb), (eps, minpts)
        ref_l, ref_n = O.dbscan_batch(xy, off[:-1], np.diff(off).astype(np.uint32), eps, minpts)
        assert np.array_equal(la, ref_l) and np.array_equal(na, ref_n), (eps, minpts)
