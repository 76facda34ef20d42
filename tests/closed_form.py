"""Numpy restatement of the *data-parallel* DBSCAN formulation the HIP kernel implements
(design/03_dbscan.md): kd-cell bounds lo/hi from the insertion-order tree, the directed
neighbour relation "in ball and not pruned", and min-seed reachability labels.

Test infrastructure only: it lets the CPU suite prove that the closed form is equivalent to the
sequential oracle (oracle/dbscan_oracle.cpp) without a GPU.  O(n^2) memory — small n only.
"""
import numpy as np


def kd_bounds(xy):
    """Level-synchronous emulation of sequential kd_insert (kdtree.cpp:106-146).

    Returns int arrays lo[n,2], hi[n,2]: pid of the deepest ancestor splitting on dim d whose
    right (lo) / left (hi) subtree holds the point, or -1.
    """
    n = xy.shape[0]
    lo = np.full((n, 2), -1, dtype=np.int64)
    hi = np.full((n, 2), -1, dtype=np.int64)
    if n == 0:
        return lo, hi
    cur = np.zeros(n, dtype=np.int64)      # node each unplaced point is compared against
    depth = np.zeros(n, dtype=np.int64)
    placed = np.zeros(n, dtype=bool)
    placed[0] = True
    BIG = np.iinfo(np.int64).max
    while not placed.all():
        act = np.nonzero(~placed)[0]
        a = cur[act]
        d = depth[act] & 1
        side = (xy[act, d] >= xy[a, d]).astype(np.int64)   # 0 = left (<), 1 = right
        child = np.full(2 * n, BIG, dtype=np.int64)
        np.minimum.at(child, 2 * a + side, act)
        c = child[2 * a + side]
        right = side == 1
        lo[act[right], d[right]] = a[right]
        hi[act[~right], d[~right]] = a[~right]
        now = c == act
        placed[act[now]] = True
        mv = ~now
        cur[act[mv]] = c[mv]
        depth[act[mv]] += 1
    return lo, hi


def edges(xy, eps):
    """Boolean matrix E[i,j] = j is returned by regionQuery(i) (self excluded)."""
    n = xy.shape[0]
    x = xy[:, 0]
    y = xy[:, 1]
    dx = x[None, :] - x[:, None]            # node.pos - query.pos  (kdtree.cpp:157)
    dy = y[None, :] - y[:, None]
    d2 = dx * dx + dy * dy
    ball = d2 <= eps * eps
    lo, hi = kd_bounds(xy)
    pruned = np.zeros((n, n), dtype=bool)
    for d in range(2):
        q = xy[:, d][:, None]                # query coordinate, per row i
        lov = np.where(lo[:, d] >= 0, xy[np.maximum(lo[:, d], 0), d], -np.inf)[None, :]
        hiv = np.where(hi[:, d] >= 0, xy[np.maximum(hi[:, d], 0), d], np.inf)[None, :]
        with np.errstate(invalid="ignore"):
            # j sits in the right subtree of an ancestor with coordinate lov: the query goes
            # left first when q <= lov (dx <= 0) and only crosses if fabs(q - lov) < eps.
            pruned |= (q <= lov) & ~(np.abs(q - lov) < eps)
            pruned |= (q > hiv) & ~(np.abs(q - hiv) < eps)
    E = ball & ~pruned
    np.fill_diagonal(E, False)
    return E


def labels_from_edges(E, minpts):
    n = E.shape[0]
    core = E.sum(axis=1) >= minpts
    lab = np.where(core, np.arange(n), n).astype(np.int64)
    Ec = E & core[:, None] & core[None, :]
    while True:
        # push along directed edges i -> j
        cand = np.where(Ec, lab[:, None], n).min(axis=0) if n else lab
        new = np.minimum(lab, cand)
        new = np.where(core, new, n)
        if (new == lab).all():
            break
        lab = new
    seeds = np.nonzero(core & (lab == np.arange(n)))[0]
    rank = np.full(n + 1, -1, dtype=np.int64)
    rank[seeds] = np.arange(seeds.shape[0])
    out = np.where(core, rank[np.minimum(lab, n)], -1).astype(np.int32)
    return out, seeds.shape[0]


def dbscan(xy, eps, minpts):
    xy = np.asarray(xy, dtype=np.float64).reshape(-1, 2)
    if xy.shape[0] < 1 or minpts < 1:
        return np.full(xy.shape[0], -1, np.int32), 0
    return labels_from_edges(edges(xy, eps), minpts)
