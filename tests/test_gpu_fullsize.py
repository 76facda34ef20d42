"""GPU, BASELINE config sizes (10 M events = configs[1]): size-independent properties of the detection path,
plus oracle spot checks on a random sample of windows (bounded CPU time)."""
import numpy as np
import pytest

import oracle_lib as O
import synth_stream as SS

pytestmark = pytest.mark.gpu
N_EVENTS = 10_000_000


@pytest.fixture(scope="module")
def run():
    import torch
    import eventcalib_amd
    from eventcalib_amd.pipeline import DetectPipeline
    ctx = eventcalib_amd.Context(0)
    pipe = DetectPipeline(ctx)
    ev = SS.make_stream(N_EVENTS, device="cuda")
    t0, t1 = SS.tiled_windows(5.0, 5.0 + (N_EVENTS - 1) / 1e6)
    pipe.set_windows(t0, t1)
    pipe.run(ev)
    torch.cuda.synchronize()
    yield ctx, pipe, ev, t0, t1, torch
    ctx.close()


def test_structure_invariants(run):
    ctx, pipe, ev, t0, t1, torch = run
    S = len(t0)
    assert not pipe.overflowed()
    lo, hi, base = pipe.win_lo[:S].long(), pipe.win_hi[:S].long(), pipe.win_base[:S + 1].long()
    # tiled windows: every event in exactly one window, in order
    assert int(lo[0]) == 0 and int(hi[-1]) == N_EVENTS and bool((lo[1:] == hi[:-1]).all())
    assert bool((base[1:] - base[:-1] == hi - lo).all()) and int(base[-1]) == N_EVENTS
    off, cnt = pipe.seg_off[:2 * S].long(), pipe.seg_cnt[:2 * S].long()
    assert bool((off[0::2] == base[:-1]).all()) and bool((off[1::2] == off[0::2] + cnt[0::2]).all())
    assert bool((cnt[0::2] + cnt[1::2] <= hi - lo).all())
    # labels: -1 or in [0, n_clusters) of their segment, and every cluster id is used
    seg_of = torch.repeat_interleave(torch.arange(2 * S, device="cuda"), cnt)
    slot = torch.repeat_interleave(off, cnt) + (torch.arange(int(cnt.sum()), device="cuda") -
                                               torch.repeat_interleave(torch.cumsum(cnt, 0) - cnt, cnt))
    lab = pipe.labels[slot].long()
    ncl = pipe.n_clusters[:2 * S].long()
    assert bool((lab >= -1).all()) and bool((lab < ncl[seg_of]).all())
    mx = torch.full((2 * S,), -1, dtype=torch.long, device="cuda").scatter_reduce(0, seg_of, lab, reduce="amax")
    assert bool((mx == ncl - 1).all())
    # a cluster's id is the rank of its smallest pid: first occurrence of each label increases with the label
    pid = slot - off[seg_of]
    key = seg_of * 4096 + lab.clamp(min=0)
    first = torch.full((2 * S * 4096,), 1 << 30, dtype=torch.long, device="cuda").scatter_reduce(
        0, key[lab >= 0], pid[lab >= 0], reduce="amin").view(2 * S, 4096)
    valid = torch.arange(4096, device="cuda")[None, :] < ncl[:, None]
    inc = (first[:, 1:] > first[:, :-1]) | ~valid[:, 1:]
    assert bool(inc.all())
    # event -> point map: -1 or a valid index of its polarity's segment
    ep = pipe.event_point[:N_EVENTS].long()
    pol = ev.view(-1, 25)[:, 24].long()
    win = torch.repeat_interleave(torch.arange(S, device="cuda"), hi - lo)
    lim = torch.where(pol > 0, cnt[2 * win], cnt[2 * win + 1])
    assert bool((ep >= -1).all()) and bool((ep < lim).all())
    # kept labels are a renumbering of labels; candidates never exceed kept + clusters
    info = pipe.win_info[:S].long()
    assert bool((info[:, 0] <= info[:, 1]).all()) and bool(((info[:, 3] == 0) | (info[:, 0] == 0)).all())


def test_partition_property(run):
    """Processing the two halves of the window list separately gives the same per-window results
    (no cross-window state): compare labels / candidates of windows in the second half."""
    ctx, pipe, ev, t0, t1, torch = run
    from eventcalib_amd.pipeline import DetectPipeline
    S = len(t0)
    h = S // 2
    p2 = DetectPipeline(ctx)
    p2.set_windows(t0[h:], t1[h:])
    p2.run(ev, slots=N_EVENTS)
    torch.cuda.synchronize()
    S2 = S - h
    assert torch.equal(p2.seg_cnt[:2 * S2], pipe.seg_cnt[2 * h:2 * S])
    assert torch.equal(p2.n_clusters[:2 * S2], pipe.n_clusters[2 * h:2 * S])
    assert torch.equal(p2.win_info[:S2], pipe.win_info[h:S])
    b0 = int(pipe.win_base[h])
    m = int(p2.win_base[S2])
    assert m == N_EVENTS - b0
    # per-slot arrays agree where slots are defined (positive+negative points of each window)
    cnt = p2.seg_cnt[:2 * S2].long()
    off = p2.seg_off[:2 * S2].long()
    slot = torch.repeat_interleave(off, cnt) + (torch.arange(int(cnt.sum()), device="cuda") -
                                               torch.repeat_interleave(torch.cumsum(cnt, 0) - cnt, cnt))
    assert torch.equal(p2.labels[slot], pipe.labels[slot + b0])
    assert torch.equal(p2.kept_labels[slot], pipe.kept_labels[slot + b0])
    assert torch.equal(p2.xy[slot], pipe.xy[slot + b0])


def test_oracle_spot_checks(run):
    """`.bin`-level parity on a sample of windows: the records of a window -> the reference's point order (real
    std::unordered_set + EigenMatrixHash, EventFrame.cpp:10-36) -> DBSCAN::Run on the reference's own kd-tree
    (oracle/_ref, dbscan.h:115-265) == the pipeline's points and labels, bit for bit."""
    ctx, pipe, ev, t0, t1, torch = run
    assert ctx.point_order() == "reference"
    S = len(t0)
    rng = np.random.default_rng(0)
    pick = np.sort(rng.choice(S, 40, replace=False))
    lo = pipe.win_lo[:S].cpu().numpy().astype(np.int64)
    hi = pipe.win_hi[:S].cpu().numpy().astype(np.int64)
    off = pipe.seg_off[:2 * S].cpu().numpy().astype(np.int64)
    cnt = pipe.seg_cnt[:2 * S].cpu().numpy().astype(np.int64)
    for s in pick:
        rec = ev[25 * lo[s]: 25 * hi[s]].cpu().numpy()
        pos, neg, ep = O.event_frame(rec, 0, hi[s] - lo[s], "reference")
        assert np.array_equal(pipe.event_point[pipe.win_base[s]:pipe.win_base[s] + (hi[s] - lo[s])].cpu().numpy(), ep)
        for k, pts in ((0, pos), (1, neg)):
            o, n = off[2 * s + k], cnt[2 * s + k]
            assert n == pts.shape[0]
            assert np.array_equal(pipe.xy[o:o + n].cpu().numpy(), pts)
            rc, lab, nc = O.dbscan(pts, 4.0, 2)
            assert np.array_equal(pipe.labels[o:o + n].cpu().numpy(), lab)


def test_every_window_of_the_10M_stream_equals_the_oracle(run):
    """BASELINE configs[1], no sampling: all 6 667 windows — bounds, points in the reference's order, event->point map,
    DBSCAN labels, cluster counts, kept labels, verdicts, representatives (the reference's nth_element picks), pairs and
    circles — against the oracle's whole-stream loop on all host threads.  Run twice when the reference's compiled
    kdtree.cpp is present (oracle/_ref): on it and on the restated tree, which therefore agree with each other too."""
    import full_compare as FC
    ctx, pipe, ev, t0, t1, torch = run
    rec = ev.cpu().numpy()
    backends = [True, False] if O.have_ref_kdtree() else [False]
    try:
        for ref in backends:
            O.set_kd_backend(ref)
            st = FC.compare_all_windows(pipe, rec, t0, t1, torch)
            print("\n[parity] 10 M events: %d windows, %d points, %d paired windows (%d with tied medians), %d candidates == oracle on %s "
                  "(%d threads, %.1f s)" % (st["windows"], st["points"], st["paired"], st["tied"], st["candidates"], st["kd_backend"],
                                           st["threads"], st["oracle_seconds"]))
            assert st["events"] == N_EVENTS and st["windows"] == len(t0)
            assert st["paired"] > len(t0) // 2 and st["tied"] > len(t0) // 10
    finally:
        FC.use_reference_kdtree_if_present()


def test_pixel_kernel_equals_general_tiers_repeatedly(run):
    """The pixel DBSCAN kernel against the general tiers (ECAL_FORCE=dbscan_general) on every segment of the stream, eight
    runs: 13 k segments x 8 is what it takes to see a one-in-10^5 ordering race between workgroup threads (the
    flatten pass once lost a root to a concurrent path-halving store)."""
    import os
    ctx, pipe, ev, t0, t1, torch = run
    from eventcalib_amd.pipeline import DetectPipeline
    S = len(t0)
    p2 = DetectPipeline(ctx)
    p2.set_windows(t0, t1)
    os.environ["ECAL_FORCE"] = "dbscan_general"
    __import__("eventcalib_amd.capi", fromlist=["sync_env"]).sync_env()   # (the switches are read once per context)
    try:
        p2.run(ev, detect=False)
        torch.cuda.synchronize()
    finally:
        os.environ.pop("ECAL_FORCE", None)
        __import__("eventcalib_amd.capi", fromlist=["sync_env"]).sync_env()   # (the switches are read once per context)
    ref_l, ref_n = p2.labels.clone(), p2.n_clusters[:2 * S].clone()
    off, cnt = p2.seg_off[:2 * S].long(), p2.seg_cnt[:2 * S].long()
    used = torch.repeat_interleave(off, cnt) + (torch.arange(int(cnt.sum()), device="cuda") -
                                               torch.repeat_interleave(torch.cumsum(cnt, 0) - cnt, cnt))
    for rep in range(8):
        p2.labels.fill_(-7)
        p2.run(ev, detect=False)
        torch.cuda.synchronize()
        assert torch.equal(p2.n_clusters[:2 * S], ref_n), rep
        assert torch.equal(p2.labels[used], ref_l[used]), rep


def test_config0_one_window_of_100k_events():
    """BASELINE configs[0]: a 100 k-event stream taken as ONE time slice (StartTime/EndTime cover everything) — the window
    goes through the global-scratch tiers (slice_big_kernel, dbscan_big_kernel, the global extraction path) instead of the
    per-window LDS kernels.  Points in the reference's order, labels on the reference's kd-tree, candidates: == oracle."""
    import torch
    import eventcalib_amd
    from eventcalib_amd.pipeline import DetectPipeline
    ctx = eventcalib_amd.Context(0)
    try:
        pipe = DetectPipeline(ctx)
        n = 100_000
        buf = SS.make_stream(n, device="cpu", seed=3)
        rec = buf.numpy()
        t, _, _ = SS.unpack_records(buf)
        pipe.set_windows([float(t[0])], [float(t[-1])])
        pipe.set_detect_params(5, 36, 15.511363636363637)
        pipe.run(buf.cuda())
        torch.cuda.synchronize()
        assert not pipe.overflowed()
        assert int(pipe.win_lo[0]) == 0 and int(pipe.win_hi[0]) == n
        pos, neg, ep = O.event_frame(rec, 0, n, "reference")
        cnt = pipe.seg_cnt[:2].cpu().numpy()
        off = pipe.seg_off[:2].cpu().numpy()
        assert (int(cnt[0]), int(cnt[1])) == (len(pos), len(neg)) and len(pos) > 4096 and len(neg) > 4096   # beyond every LDS tier
        assert np.array_equal(pipe.xy[off[0]:off[0] + cnt[0]].cpu().numpy(), pos)
        assert np.array_equal(pipe.xy[off[1]:off[1] + cnt[1]].cpu().numpy(), neg)
        assert np.array_equal(pipe.event_point[:n].cpu().numpy(), ep)
        for k, pts in ((0, pos), (1, neg)):
            rc, lab, nc = O.dbscan(pts, 4.0, 2)
            assert np.array_equal(pipe.labels[off[k]:off[k] + cnt[k]].cpu().numpy(), lab), k
            assert int(pipe.n_clusters[k]) == nc
        ref = O.extract_candidates(pos, neg, 4.0, 2, 5, 36, 15.511363636363637)
        info = pipe.win_info[0].cpu().numpy()
        assert info[3] == ref["status"] and info[1] == ref["nk_pos"] and info[2] == ref["nk_neg"]
        assert np.array_equal(pipe.kept_labels[off[0]:off[0] + cnt[0]].cpu().numpy(), ref["kept_pos"])
        assert np.array_equal(pipe.kept_labels[off[1]:off[1] + cnt[1]].cpu().numpy(), ref["kept_neg"])
        # representatives (the reference's nth_element picks: the member order of these > 4096-point segments comes from the
        # global-scratch launch of ecal_cluster_order_dev), pairs, circles — and no fallback flag on the window
        assert int(info[3]) & 0x100 == 0
        if not ref["status"]:
            assert np.array_equal(pipe.rep[off[0]:off[0] + ref["nk_pos"]].cpu().numpy(), ref["rep_pos"])
            assert np.array_equal(pipe.rep[off[1]:off[1] + ref["nk_neg"]].cpu().numpy(), ref["rep_neg"])
            nc = ref["n"]
            assert int(info[0]) == nc
            assert np.array_equal(pipe.cand_pair[off[0]:off[0] + nc].cpu().numpy(), ref["pair"])
            assert np.array_equal(pipe.cand_xyr[off[0]:off[0] + nc].cpu().numpy(), ref["xyr"])
    finally:
        ctx.close()


def test_a_tie_pick_that_cannot_be_reproduced_is_flagged_not_silent(monkeypatch):
    """The one case left in which the exact extraction falls back to the plain rule (smaller pid among the tied medians): a
    segment beyond the member-order kernel's global workspace — forced here by shrinking its range-query arena
    (ECAL_BO_BIG_ARENA, a test switch) below what the tied clusters' range queries need.  The window must say so
    (ECAL_WIN_TIE_FALLBACK in win_info[..][3]); every representative is then either the reference's pick (the polarity whose
    queries still fitted) or the plain primitive's, and the rest of the window follows from them as usual."""
    import torch
    import eventcalib_amd
    from eventcalib_amd.pipeline import DetectPipeline
    n = 100_000
    buf = SS.make_stream(n, device="cpu", seed=3)
    t, _, _ = SS.unpack_records(buf)
    outs = {}
    for mode in ("exact", "exact_small_arena", "plain"):
        if mode == "exact_small_arena":
            monkeypatch.setenv("ECAL_BO_BIG_ARENA", "64")
            __import__("eventcalib_amd.capi", fromlist=["sync_env"]).sync_env()   # (the switches are read once per context)
        else:
            monkeypatch.delenv("ECAL_BO_BIG_ARENA", raising=False)
            __import__("eventcalib_amd.capi", fromlist=["sync_env"]).sync_env()   # (the switches are read once per context)
        ctx = eventcalib_amd.Context(0)
        try:
            pipe = DetectPipeline(ctx)
            pipe.set_windows([float(t[0])], [float(t[-1])])
            pipe.set_detect_params(5, 36, 15.511363636363637)
            pipe.run(buf.cuda(), exact_ties=mode != "plain")
            torch.cuda.synchronize()
            off = pipe.seg_off[:2].cpu().numpy()
            info = pipe.win_info[0].cpu().numpy().astype(np.int64)
            outs[mode] = dict(info=info, rep=np.concatenate([pipe.rep[off[0]:off[0] + info[1]].cpu().numpy(),
                                                              pipe.rep[off[1]:off[1] + info[2]].cpu().numpy()]),
                              pair=pipe.cand_pair[off[0]:off[0] + info[0]].cpu().numpy(),
                              xyr=pipe.cand_xyr[off[0]:off[0] + info[0]].cpu().numpy())
        finally:
            ctx.close()
    e, a, b = outs["exact"], outs["exact_small_arena"], outs["plain"]
    assert int(a["info"][3]) & 0x100, "the fallback must be reported"
    assert int(b["info"][3]) & 0x100 == 0 and int(e["info"][3]) & 0x100 == 0
    assert (int(a["info"][3]) & 0xFF) == int(b["info"][3]) == int(e["info"][3])
    assert np.array_equal(a["info"][1:3], b["info"][1:3]) and np.array_equal(a["info"][1:3], e["info"][1:3])
    differ = e["rep"] != b["rep"]
    assert differ.any()                                                     # (the window does hold medians the two rules pick differently)
    assert np.all((a["rep"] == e["rep"]) | (a["rep"] == b["rep"]))
    assert np.any(differ & (a["rep"] == b["rep"]))                           # some pick really is the plain rule's
    for other in (e, b):   # with the same representatives everything downstream is the same, bit for bit
        if np.array_equal(a["rep"], other["rep"]):
            assert a["info"][0] == other["info"][0]
            assert np.array_equal(a["pair"], other["pair"]) and np.array_equal(a["xyr"], other["xyr"])


@pytest.mark.parametrize("n_events", [50_000_000])
def test_every_window_at_the_benchmark_size_equals_the_oracle(n_events):
    """BASELINE configs[2] (the 50 M-event stream bench.py times): ALL 33 334 windows, records -> reference point order
    (real std::unordered_set) -> labels on the reference's compiled kd-tree (oracle/_ref, when present) -> kept clusters ->
    the reference's nth_element picks -> pairs and circles, == the oracle in every slot of every array; plus the structure
    invariants that do not need the oracle.  ~9 GB of HBM, ~4 GB of host memory, seconds on the box's host threads."""
    import torch
    import eventcalib_amd
    import full_compare as FC
    from eventcalib_amd.pipeline import DetectPipeline
    if torch.cuda.get_device_properties(0).total_memory < 40e9:
        pytest.skip("needs ~10 GB of device memory")
    ctx = eventcalib_amd.Context(0)
    try:
        pipe = DetectPipeline(ctx)
        ev = SS.make_stream(n_events, device="cuda")
        t0, t1 = SS.tiled_windows(5.0, 5.0 + (n_events - 1) / 1e6)
        S = len(t0)
        pipe.set_windows(t0, t1)
        pipe.set_detect_params(5, 36, 15.511363636363637)
        pipe.run(ev)
        torch.cuda.synchronize()
        assert not pipe.overflowed()
        lo, hi = pipe.win_lo[:S].long(), pipe.win_hi[:S].long()
        assert int(lo[0]) == 0 and int(hi[-1]) == n_events and bool((lo[1:] == hi[:-1]).all())
        cnt = pipe.seg_cnt[:2 * S].long()
        assert bool((cnt[0::2] + cnt[1::2] <= hi - lo).all())
        backend = FC.use_reference_kdtree_if_present()
        st = FC.compare_all_windows(pipe, ev.cpu().numpy(), t0, t1, torch)
        print("\n[parity] 50 M events: %d windows, %d points, %d paired windows (%d with tied medians), %d candidates == oracle on %s "
              "(%d threads, %.1f s)" % (st["windows"], st["points"], st["paired"], st["tied"], st["candidates"], backend, st["threads"],
                                       st["oracle_seconds"]))
        assert st["events"] == n_events and st["windows"] == S == 33334
        assert st["paired"] > S // 2 and st["tied"] > S // 10, st
    finally:
        ctx.close()
