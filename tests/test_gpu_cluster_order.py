"""GPU: ecal_cluster_order_dev — the member ORDER of the reference's Clusters[c] (expandCluster's pop order over the kd-tree's
result lists, dbscan.h:229-265, kdtree.cpp:148-179,469-486) — against the oracle driver running on the REFERENCE's own
kd-tree (oracle/_ref, where built) or on the oracle's restatement of it."""
import numpy as np
import pytest

import oracle_lib as O
import synth_stream as SS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    import eventcalib_amd
    ctx = eventcalib_amd.Context(0)
    yield ctx, torch
    ctx.close()


def _order_of(ctx, torch, segs, eps, minpts):
    """segs: list of [n][2] float64 arrays -> (labels, order, status) per segment through the C ABI."""
    S = len(segs)
    cnt = np.array([len(s) for s in segs], dtype=np.int32)
    off = np.concatenate([[0], np.cumsum(cnt)[:-1]]).astype(np.int32)
    N = int(cnt.sum())
    xy = torch.as_tensor(np.concatenate(segs).reshape(-1, 2) if N else np.zeros((0, 2)), device="cuda").contiguous()
    d_off, d_cnt = torch.as_tensor(off, device="cuda"), torch.as_tensor(cnt, device="cuda")
    labels = torch.empty(max(N, 1), dtype=torch.int32, device="cuda")
    ncl = torch.empty(S, dtype=torch.int32, device="cuda")
    order = torch.full((max(N, 1),), -7, dtype=torch.int32, device="cuda")
    status = torch.full((S,), -7, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    ctx.dbscan_batch_dev(xy.data_ptr(), d_off.data_ptr(), d_cnt.data_ptr(), S, N, int(cnt.max()) if S else 0, eps, minpts,
                         labels.data_ptr(), ncl.data_ptr(), st)
    ctx.cluster_order_dev(xy.data_ptr(), d_off.data_ptr(), d_cnt.data_ptr(), S, eps, labels.data_ptr(), ncl.data_ptr(),
                          order.data_ptr(), status.data_ptr(), st)
    torch.cuda.synchronize()
    lab, od, stt = labels.cpu().numpy(), order.cpu().numpy(), status.cpu().numpy()
    return [(lab[off[s]:off[s] + cnt[s]], od[off[s]:off[s] + cnt[s]], int(stt[s])) for s in range(S)]


def _check(seg, lab, od, eps, minpts):
    # the oracle's k-d tree is the session's backend (tests/conftest.py: the reference's compiled kdtree.cpp when present)
    _rc, labels_o, _nc, clusters = O.dbscan(seg, eps, minpts, with_members=True)
    assert np.array_equal(lab, labels_o)
    assert np.array_equal(od < 0, lab < 0)
    for c, members in enumerate(clusters):
        idx = np.flatnonzero(lab == c)
        mine = idx[np.argsort(od[idx], kind="stable")]
        assert np.array_equal(np.sort(od[idx]), np.arange(len(idx))), c          # positions 0 .. size - 1, each once
        assert np.array_equal(mine, members), (c, mine[:12], members[:12])


def test_stream_windows_match_the_reference_member_order(env):
    ctx, torch = env
    from eventcalib_amd.pipeline import DetectPipeline
    n = 300_000
    ev = SS.make_stream(n, device="cuda", seed=17)
    t0, t1 = SS.tiled_windows(5.0, 5.0 + (n - 1) / 1e6)
    pipe = DetectPipeline(ctx)
    pipe.set_windows(t0, t1)
    pipe.run(ev, slice_only=True)
    torch.cuda.synchronize()
    S = len(t0)
    off, cnt = pipe.seg_off[:2 * S].cpu().numpy(), pipe.seg_cnt[:2 * S].cpu().numpy()
    xy = pipe.xy.cpu().numpy()
    segs = [xy[off[s]:off[s] + cnt[s]].copy() for s in range(0, 2 * S, 7)]
    res = _order_of(ctx, torch, segs, 4.0, 2)
    placed = 0
    for seg, (lab, od, st) in zip(segs, res):
        assert st == 0
        _check(seg, lab, od, 4.0, 2)
        placed += int((od > 0).sum())
    assert placed > 5000          # the order is not trivial: thousands of non-seed members were placed


@pytest.mark.parametrize("eps,minpts", [(4.0, 2), (3.0, 3), (4.5, 2), (2.5, 1), (1.7, 5)])
def test_random_segments(env, eps, minpts):
    """Integer lattices (the pruning quirk bites at integral eps), half-pixel lattices and continuous coordinates, 1 .. 2048 points."""
    ctx, torch = env
    rng = np.random.default_rng(int(eps * 10) + minpts)
    segs = []
    for n in (1, 2, 3, 17, 64, 200, 650, 1024, 2048):
        side = max(4, int(np.sqrt(n) * 2.2))
        pts = rng.permutation(side * side)[:n]
        segs.append(np.stack([pts % side, pts // side], 1).astype(np.float64))                 # unique integer pixels
        segs.append(np.stack([pts % side, pts // side], 1).astype(np.float64) * 0.5 + 3.25)    # half-pixel lattice
        segs.append(rng.uniform(0, side, size=(n, 2)))                                        # continuous
    res = _order_of(ctx, torch, segs, eps, minpts)
    for seg, (lab, od, st) in zip(segs, res):
        assert st == 0     # (eps-balls of more than 64 other points — dense lattices at the larger radii — go to the global-scratch launch)
        _check(seg, lab, od, eps, minpts)


def test_oversized_segments_and_degenerate_trees_are_taken(env):
    """Beyond the LDS tiers (> 4096 points, > 2048 clusters, > 64 hits per eps-ball, > 96 pending subtrees) the global-scratch
    launch takes the segment: stackless walks in result-list order, no fixed list sizes."""
    ctx, torch = env
    rng = np.random.default_rng(5)
    big = np.stack([np.arange(5000) % 80, np.arange(5000) // 80], 1).astype(np.float64)[rng.permutation(5000)]   # > 4096 points
    big2 = np.stack([np.arange(20000) % 200, np.arange(20000) // 200], 1).astype(np.float64)[rng.permutation(20000)]
    sparse = rng.uniform(0, 2000, size=(9000, 2))                               # thousands of small clusters and noise
    chain = np.stack([np.arange(400, dtype=np.float64), np.zeros(400)], 1)      # sorted insertion: a tree 400 levels deep
    zig = np.stack([np.arange(300, dtype=np.float64) * 0.01, (np.arange(300) % 2) * 0.01], 1)   # deep AND every far subtree pending
    ok = rng.uniform(0, 30, size=(300, 2))
    blob = rng.uniform(0, 6, size=(500, 2))                                     # hundreds of hits per eps-ball
    segs = [big, chain, ok, big2, sparse, zig, blob]
    res = _order_of(ctx, torch, segs, 4.0, 2)
    for seg, r in zip(segs, res):
        assert r[2] == 0
        _check(seg, r[0], r[1], 4.0, 2)


def test_tier_boundaries_of_the_two_launches(env):
    """The first launch takes <= 768 points and <= 256 clusters per segment, the second up to 2048, the third up to 4096 points: same answers."""
    ctx, torch = env
    rng = np.random.default_rng(21)
    segs = []
    for pairs in (250, 256, 257, 380):                       # clusters of two pixels, 10 apart: n = 2 * pairs <= 768, n_clusters = pairs
        cells = rng.permutation(40 * 40)[:pairs]
        a = np.stack([cells % 40, cells // 40], 1).astype(np.float64) * 10.0
        pts = np.concatenate([a, a + [1.0, 0.0]])
        segs.append(pts[rng.permutation(len(pts))])
    for n in (767, 768, 769, 900, 2048, 2049, 3000, 4096):
        side = int(np.sqrt(n) * 2.2)
        pts = rng.permutation(side * side)[:n]
        segs.append(np.stack([pts % side, pts // side], 1).astype(np.float64))
    res = _order_of(ctx, torch, segs, 4.0, 1)                # (minPts 1: one neighbour makes a core point)
    for seg, (lab, od, st) in zip(segs, res):
        assert st == 0
        _check(seg, lab, od, 4.0, 1)
    assert [int(r[0].max()) + 1 for r in res[:4]] == [250, 256, 257, 380]


def test_host_buffer_form_equals_the_device_form(env):
    """ecal_cluster_order (what the DBSCAN<T,Float> shim calls to put Clusters[c] into the reference's order)."""
    ctx, torch = env
    rng = np.random.default_rng(8)
    segs = [rng.integers(0, 40, size=(n, 2)).astype(np.float64) for n in (300, 1, 0, 700)]
    segs = [np.unique(s, axis=0)[rng.permutation(len(np.unique(s, axis=0)))] if len(s) else s for s in segs]
    res = _order_of(ctx, torch, segs, 4.0, 2)
    off = np.concatenate([[0], np.cumsum([len(s) for s in segs])]).astype(np.uint32)
    xy = np.concatenate([s.reshape(-1, 2) for s in segs])
    labels = np.concatenate([r[0] for r in res])
    ncl = np.array([int(r[0].max()) + 1 if len(r[0]) and r[0].max() >= 0 else 0 for r in res], np.uint32)
    order, status = ctx.cluster_order(xy, off, 4.0, labels, ncl)
    assert np.array_equal(status, [r[2] for r in res])
    assert np.array_equal(order, np.concatenate([r[1] for r in res]))


def test_only_tied_medians_mode(env):
    """only_tied_medians: the same positions for the clusters whose median has an equal-norm rival, -2 for all other members."""
    ctx, torch = env
    from eventcalib_amd.pipeline import DetectPipeline
    n = 400_000
    ev = SS.make_stream(n, device="cuda", seed=23)
    t0, t1 = SS.tiled_windows(5.0, 5.0 + (n - 1) / 1e6)
    pipe = DetectPipeline(ctx)
    pipe.set_windows(t0, t1)
    pipe.run(ev, detect=False)
    S = len(t0)
    st = torch.cuda.current_stream().cuda_stream
    full = torch.empty(n, dtype=torch.int32, device="cuda")
    part = torch.full((n,), -9, dtype=torch.int32, device="cuda")
    s1 = torch.empty(2 * S, dtype=torch.int32, device="cuda")
    s2 = torch.empty(2 * S, dtype=torch.int32, device="cuda")
    args = (pipe.xy.data_ptr(), pipe.seg_off.data_ptr(), pipe.seg_cnt.data_ptr(), 2 * S, 4.0, pipe.labels.data_ptr(), pipe.n_clusters.data_ptr())
    ctx.cluster_order_dev(*args, full.data_ptr(), s1.data_ptr(), st)
    ctx.cluster_order_dev(*args, part.data_ptr(), s2.data_ptr(), st, only_tied_medians=True)
    torch.cuda.synchronize()
    assert int(s1.sum()) == 0 and int(s2.sum()) == 0
    off, cnt = pipe.seg_off[:2 * S].cpu().numpy().astype(np.int64), pipe.seg_cnt[:2 * S].cpu().numpy().astype(np.int64)
    full, part = full.cpu().numpy(), part.cpu().numpy()
    lab, xy = pipe.labels.cpu().numpy(), pipe.xy.cpu().numpy()
    tied_clusters = plain_clusters = 0
    for sg in range(2 * S):
        o, m = off[sg], cnt[sg]
        f, p, l, pts = full[o:o + m], part[o:o + m], lab[o:o + m], xy[o:o + m]
        assert np.array_equal(p == -1, l < 0)
        key = (pts ** 2).sum(1)
        for c in range(int(l.max()) + 1 if m else 0):
            idx = np.flatnonzero(l == c)
            srt = idx[np.lexsort((idx, key[idx]))]                    # (norm, pid)
            mid = srt[len(idx) // 2]
            tied = int((key[idx] == key[mid]).sum()) > 1
            if tied:
                tied_clusters += 1
                assert np.array_equal(p[idx], f[idx]), (sg, c)
            else:
                plain_clusters += 1
                assert (p[idx] == -2).all(), (sg, c)
    assert tied_clusters > 10 and plain_clusters > 20 * tied_clusters


def test_marked_clusters_mode(env):
    """only_tied_medians = 2 (what ecal_extract_batch_exact_dev uses): the caller names the clusters by a -3 on the slot of one
    member each; those get their positions, every other cluster -2, segments without a mark are left without a tree."""
    ctx, torch = env
    rng = np.random.default_rng(31)
    segs = [rng.integers(0, 45, size=(n, 2)).astype(np.float64) for n in (500, 300, 760, 1500)]
    segs = [np.unique(s, axis=0)[rng.permutation(len(np.unique(s, axis=0)))] for s in segs]
    res = _order_of(ctx, torch, segs, 4.0, 2)
    cnt = np.array([len(s) for s in segs], dtype=np.int32)
    off = np.concatenate([[0], np.cumsum(cnt)[:-1]]).astype(np.int32)
    labels = np.concatenate([r[0] for r in res]).astype(np.int32)
    full = np.concatenate([r[1] for r in res])
    ncl = np.array([int(r[0].max()) + 1 for r in res], dtype=np.int32)
    marks = np.full(len(labels), 7, dtype=np.int32)          # stale content of the buffer: anything but the mark
    wanted = []
    for s in (0, 2, 3):                                      # segment 1 stays unmarked
        ids = rng.permutation(ncl[s])[: max(1, ncl[s] // 3)]
        for c in ids:
            members = np.flatnonzero(labels[off[s]:off[s] + cnt[s]] == c)
            marks[off[s] + members[rng.integers(len(members))]] = -3
            wanted.append((s, int(c)))
    xy = torch.as_tensor(np.concatenate(segs), device="cuda").contiguous()
    d = lambda a: torch.as_tensor(a, device="cuda")
    d_off, d_cnt, d_lab, d_ncl, d_ord = d(off), d(cnt), d(labels), d(ncl), d(marks)
    status = torch.full((4,), -7, dtype=torch.int32, device="cuda")
    ctx.cluster_order_dev(xy.data_ptr(), d_off.data_ptr(), d_cnt.data_ptr(), 4, 4.0, d_lab.data_ptr(), d_ncl.data_ptr(), d_ord.data_ptr(),
                          status.data_ptr(), torch.cuda.current_stream().cuda_stream, only_tied_medians=2)
    torch.cuda.synchronize()
    got = d_ord.cpu().numpy()
    assert status.cpu().tolist() == [0, 0, 0, 0]
    want = np.where(labels < 0, -1, -2).astype(np.int32)
    for s, c in wanted:
        sel = off[s] + np.flatnonzero(labels[off[s]:off[s] + cnt[s]] == c)
        want[sel] = full[sel]
    assert np.array_equal(got, want)
