"""GPU parity: candidate extraction (through the C ABI) vs the oracle."""
import numpy as np
import pytest

import oracle_lib as O
import synth_stream as SS

pytestmark = pytest.mark.gpu
THR = 15.511363636363637


@pytest.fixture(scope="module")
def env():
    import torch
    import eventcalib_amd
    from eventcalib_amd.pipeline import DetectPipeline
    ctx = eventcalib_amd.Context(0)
    yield ctx, DetectPipeline(ctx), torch
    ctx.close()


def test_radius_threshold_kat(env):
    ctx, pipe, torch = env
    assert ctx.circle_radius_threshold(346, 260, 9, 4, True, 5.5, 1.75) == THR
    assert ctx.circle_radius_threshold(346, 260, 9, 4, True, 5.5, 1.75) == O.circle_radius_threshold(346.0, 260.0, 9, 4,
                                                                                                    True, 5.5, 1.75)
    assert ctx.circle_radius_threshold(640, 480, 5, 7, False, 3.0, 1.0) == O.circle_radius_threshold(640.0, 480.0, 5, 7,
                                                                                                     False, 3.0, 1.0)


def _check_windows(pipe, torch, rec, t0, t1, cluster_min, need, thr, fit_circle=False, knn_num=3, exact_ties=True):
    S = len(t0)
    info = pipe.win_info[:S].cpu().numpy().astype(np.int64)
    seg_off = pipe.seg_off[:2 * S].cpu().numpy().astype(np.int64)
    seg_cnt = pipe.seg_cnt[:2 * S].cpu().numpy().astype(np.int64)
    xy = pipe.xy.cpu().numpy()
    kept = pipe.kept_labels.cpu().numpy()
    rep = pipe.rep.cpu().numpy()
    pair = pipe.cand_pair.cpu().numpy()
    xyr = pipe.cand_xyr.cpu().numpy()
    exact = tied = 0
    for s in range(S):
        op, on = seg_off[2 * s], seg_off[2 * s + 1]
        pos = xy[op:op + seg_cnt[2 * s]]
        neg = xy[on:on + seg_cnt[2 * s + 1]]
        ref = O.extract_candidates(pos, neg, 4.0, 2, cluster_min, need, thr, fit_circle, knn_num)
        assert info[s, 3] == ref["status"], "window %d status" % s
        assert np.array_equal(kept[op:op + len(pos)], ref["kept_pos"]), "window %d kept +" % s
        assert np.array_equal(kept[on:on + len(neg)], ref["kept_neg"]), "window %d kept -" % s
        if len(pos) and len(neg):
            assert info[s, 1] == ref["nk_pos"] and info[s, 2] == ref["nk_neg"]
        if ref["status"]:
            assert info[s, 0] == 0
            continue
        if ref["tie"] and exact_ties:
            tied += 1        # the default (exact ties): the reference's own pick — nothing is handed to the oracle, see below
        elif ref["tie"]:
            # Some cluster's median has an equal-norm rival: which of the two the reference's nth_element returns depends
            # on its BFS member order, the build takes the smaller pid.  That choice is the ONLY freedom:
            # every untied cluster's representative must be the oracle's, a tied cluster's must be a member of the same
            # cluster with the oracle's norm, and with those representatives handed to the oracle everything downstream
            # (pairs, circles, count) must again be identical.
            tied += 1
            over = []
            for k, (o, pts, keptl) in enumerate(((op, pos, ref["kept_pos"]), (on, neg, ref["kept_neg"]))):
                nk = ref["nk_pos"] if k == 0 else ref["nk_neg"]
                g, r, tie_c = rep[o:o + nk].astype(np.int64), ref["rep_pos" if k == 0 else "rep_neg"].astype(np.int64), ref["tie_pos" if k == 0 else "tie_neg"]
                assert np.array_equal(g[~tie_c], r[~tie_c]), "window %d untied representatives (%s)" % (s, "+-"[k])
                assert np.array_equal(keptl[g], np.arange(nk)), "window %d tied representative outside its cluster" % s
                assert np.array_equal((pts[g] ** 2).sum(1), (pts[r] ** 2).sum(1)), "window %d tied representative norm" % s
                over.append(np.where(tie_c, g, 0xFFFFFFFF).astype(np.uint32))
            ref = O.extract_candidates(pos, neg, 4.0, 2, cluster_min, need, thr, fit_circle, knn_num, over[0], over[1])
        else:
            exact += 1
        assert np.array_equal(rep[op:op + ref["nk_pos"]], ref["rep_pos"]), "window %d rep +" % s
        assert np.array_equal(rep[on:on + ref["nk_neg"]], ref["rep_neg"]), "window %d rep -" % s
        n = ref["n"]
        assert info[s, 0] == n, "window %d candidate count" % s
        assert np.array_equal(pair[op:op + n], ref["pair"]), "window %d pairs" % s
        if fit_circle:
            # the nine running sums of fitCircle are accumulated in ascending pid here and in the reference's
            # BFS member order in the oracle: f64 rounding differs in the last digits, stated tolerance 1e-9
            assert np.allclose(xyr[op:op + n], ref["xyr"], rtol=1e-9, atol=1e-9), "window %d circles" % s
        else:
            assert np.array_equal(xyr[op:op + n], ref["xyr"]), "window %d circles" % s   # same arithmetic: bitwise
    return exact, tied


@pytest.mark.parametrize("rate", [1.0e6, 2.0e6, 4.0e6])
def test_synthetic_stream(env, rate):
    ctx, pipe, torch = env
    buf = SS.make_stream(90000, rate=rate, device="cpu", seed=31)
    t, _, _ = SS.unpack_records(buf)
    t0, t1 = SS.tiled_windows(float(t[0]), float(t[-1]))
    pipe.set_windows(t0, t1)
    pipe.set_detect_params(5, 36, THR)
    pipe.run(buf.cuda())
    torch.cuda.synchronize()
    exact, tied = _check_windows(pipe, torch, buf.numpy(), t0, t1, 5, 36, THR)
    assert exact + tied > 0 or rate < 2.0e6      # sparse 1 Mev/s windows may all fail the 36-cluster test
    if rate >= 2.0e6:
        assert exact + tied >= 0.7 * len(t0), (exact, tied, len(t0))   # most windows reach the pairing stage at these rates


def test_small_need_and_cluster_min(env):
    """Looser parameters so every window reaches the pairing stage; also windows with an empty polarity."""
    ctx, pipe, torch = env
    buf = SS.make_stream(40000, rate=1.0e6, device="cpu", seed=5)
    t, xy, p = SS.unpack_records(buf)
    ts, te = float(t[0]), float(t[-1])
    t0, t1 = SS.tiled_windows(ts, te)
    t0 = list(t0) + [ts - 1.0]
    t1 = list(t1) + [ts - 0.5]
    pipe.set_windows(t0, t1)
    pipe.set_detect_params(3, 10, 40.0)
    pipe.run(buf.cuda())
    torch.cuda.synchronize()
    exact, tied = _check_windows(pipe, torch, buf.numpy(), t0, t1, 3, 10, 40.0)
    assert exact + tied >= len(t0) - 2, (exact, tied, len(t0))     # every non-empty window pairs with these parameters


def test_large_windows_use_the_global_scratch_path(env):
    """Windows with more than 1408 unique pixels do not fit the LDS staging of extract_kernel."""
    ctx, pipe, torch = env
    buf = SS.make_stream(60000, rate=2.0e6, device="cpu", seed=12)
    t, _, _ = SS.unpack_records(buf)
    t0, t1 = SS.tiled_windows(float(t[0]), float(t[-1]), 3.0e-3)
    pipe.set_windows(t0, t1)
    pipe.set_detect_params(5, 36, THR)
    pipe.run(buf.cuda())
    torch.cuda.synchronize()
    S = len(t0)
    assert int((pipe.seg_cnt[:2 * S:2] + pipe.seg_cnt[1:2 * S:2]).max()) > 1408
    exact, tied = _check_windows(pipe, torch, buf.numpy(), t0, t1, 5, 36, THR)
    assert exact + tied >= 0.7 * S, (exact, tied, S)


@pytest.mark.parametrize("knn_num", [1, 3])
def test_fit_circle_path(env, knn_num):
    """fitCircle == 1 (CirclesEventFrame.cpp:180-281): knn candidates, algebraic circle fit, double-direction check."""
    ctx, pipe, torch = env
    buf = SS.make_stream(90000, rate=2.0e6, device="cpu", seed=77)
    t, _, _ = SS.unpack_records(buf)
    t0, t1 = SS.tiled_windows(float(t[0]), float(t[-1]))
    pipe.set_windows(t0, t1)
    pipe.set_detect_params(5, 36, THR, fit_circle=True, knn_num=knn_num)
    pipe.run(buf.cuda())
    torch.cuda.synchronize()
    exact, tied = _check_windows(pipe, torch, buf.numpy(), t0, t1, 5, 36, THR, True, knn_num)
    assert exact + tied >= 0.7 * len(t0), (exact, tied, len(t0))
    pipe.set_detect_params(5, 36, THR)


@pytest.mark.parametrize("rate", [1.0e6, 2.0e6])   # first pass (<= 1408 points per window) and second pass (<= 2816)
def test_translated_pixels_take_the_plain_key_path(env, rate):
    """Coordinates beyond 1023 do not fit the composite (norm, pid) member word of extract_kernel (x^2 + y^2 must stay
    below 2^21): such windows are still staged in LDS but rank their cluster members by key and index separately,
    and search the nearest representative in doubles."""
    ctx, pipe, torch = env
    buf = SS.make_stream(60000, rate=rate, device="cpu", seed=44)
    rec = buf.numpy().reshape(-1, 25)
    xy = rec[:, 8:24].copy().view(np.float64)
    xy += np.array([3000.0, 1200.0])
    rec[:, 8:24] = xy.view(np.uint8)
    t, _, _ = SS.unpack_records(buf)
    t0, t1 = SS.tiled_windows(float(t[0]), float(t[-1]))
    pipe.set_windows(t0, t1)
    pipe.set_detect_params(5, 36, THR)
    pipe.run(buf.cuda())
    torch.cuda.synchronize()
    exact, tied = _check_windows(pipe, torch, buf.numpy(), t0, t1, 5, 36, THR)
    assert exact + tied >= 3 or rate < 2.0e6


@pytest.mark.parametrize("rate,params", [(1.0e6, (3, 10, 40.0)), (2.0e6, (5, 36, THR)), (4.0e6, (5, 36, THR))])
def test_exact_ties_reference_representatives(env, rate, params):
    """run(exact_ties=True): where a cluster's median rank has an equal-norm rival the representative is the reference's OWN pick
    (std::nth_element over Clusters[c] in expandCluster's order) — so representatives, pairs, circles and counts equal the oracle's
    in EVERY window, with nothing handed to the oracle."""
    ctx, pipe, torch = env
    buf = SS.make_stream(int(0.12 * rate), rate=rate, device="cpu", seed=41)      # 80 windows at every density
    t, _, _ = SS.unpack_records(buf)
    t0, t1 = SS.tiled_windows(float(t[0]), float(t[-1]))
    pipe.set_windows(t0, t1)
    pipe.set_detect_params(*params)
    pipe.run(buf.cuda(), exact_ties=True)
    torch.cuda.synchronize()
    exact, tied = _check_windows(pipe, torch, buf.numpy(), t0, t1, *params, exact_ties=True)
    assert tied >= 2 and exact + tied >= 0.7 * len(t0), (exact, tied, len(t0))


def test_exact_ties_in_the_global_scratch_path_and_with_fit_circle(env):
    ctx, pipe, torch = env
    buf = SS.make_stream(400000, rate=2.0e6, device="cpu", seed=12)
    t, _, _ = SS.unpack_records(buf)
    t0, t1 = SS.tiled_windows(float(t[0]), float(t[-1]), 3.0e-3)          # > 1408 unique pixels per window
    pipe.set_windows(t0, t1)
    pipe.set_detect_params(5, 36, THR)
    pipe.run(buf.cuda(), exact_ties=True)
    torch.cuda.synchronize()
    exact, tied = _check_windows(pipe, torch, buf.numpy(), t0, t1, 5, 36, THR, exact_ties=True)
    assert tied >= 1 and exact + tied >= 0.7 * len(t0)
    buf = SS.make_stream(90000, rate=2.0e6, device="cpu", seed=77)
    t, _, _ = SS.unpack_records(buf)
    t0, t1 = SS.tiled_windows(float(t[0]), float(t[-1]))
    pipe.set_windows(t0, t1)
    pipe.set_detect_params(5, 36, THR, fit_circle=True, knn_num=3)
    pipe.run(buf.cuda(), exact_ties=True)
    torch.cuda.synchronize()
    exact, tied = _check_windows(pipe, torch, buf.numpy(), t0, t1, 5, 36, THR, fit_circle=True, knn_num=3, exact_ties=True)
    assert tied >= 1
    pipe.set_detect_params(5, 36, THR)


def test_smaller_pid_rule_of_the_plain_primitive(env):
    """run(exact_ties=False) = ecal_extract_batch_dev alone: at a tied median the smaller pid — any other representative, and with
    that ONE choice handed to the oracle everything downstream, must still be identical."""
    ctx, pipe, torch = env
    buf = SS.make_stream(240000, rate=2.0e6, device="cpu", seed=43)
    t, _, _ = SS.unpack_records(buf)
    t0, t1 = SS.tiled_windows(float(t[0]), float(t[-1]))
    pipe.set_windows(t0, t1)
    pipe.set_detect_params(5, 36, THR)
    pipe.run(buf.cuda(), exact_ties=False)
    torch.cuda.synchronize()
    exact, tied = _check_windows(pipe, torch, buf.numpy(), t0, t1, 5, 36, THR, exact_ties=False)
    assert tied >= 2 and exact + tied >= 0.7 * len(t0)


def test_counter_words_from_the_ring_or_wiped_per_call(env, monkeypatch):
    """The to-do counters of the stage calls come from a per-stream ring of zeroed words (ecal_zero_words) or, without one —
    a fifth stream on one context —, from words wiped per call: same results, also across the ring's
    half-by-half wipes (600 passes = 3 600 counter words on one stream)."""
    ctx, pipe, torch = env
    n = 200_000
    ev = SS.make_stream(n, device="cuda", seed=41, rate=1.6e6)
    t0, t1 = SS.tiled_windows(5.0, 5.0 + (n - 1) / 1.6e6)
    pipe.set_windows(t0, t1)
    pipe.set_detect_params(5, 36, THR)
    S = len(t0)

    def snapshot():
        torch.cuda.synchronize()
        m = int(pipe.seg_off[2 * S - 1] + pipe.seg_cnt[2 * S - 1])
        c = int(pipe.win_info[:S, 0].sum())
        return [pipe.win_info[:S].clone(), pipe.labels[:m].clone(), pipe.kept_labels[:m].clone(), pipe.cand_pair[:c].clone(),
                pipe.cand_xyr[:c].clone()]

    pipe.run(ev)
    want = snapshot()
    for _ in range(600):                       # through several wipes of the ring
        pipe.run(ev)
    for a, b in zip(snapshot(), want):
        assert torch.equal(a, b)
    for k in range(6):                         # more streams than the context has rings
        with torch.cuda.stream(torch.cuda.Stream()):
            pipe.run(ev)
            for a, b in zip(snapshot(), want):
                assert torch.equal(a, b)



def test_records_to_candidates_with_no_gpu_array_in_between(env):
    """The whole chain on the oracle's side — packed records -> window bounds -> EventFrame (the reference's unordered_set
    order) -> extractFeatures up to the candidate circles — against ecal_detect_pass / the pipeline's arrays, with NO GPU
    array handed to the oracle (the other tests of this file feed it the sliced points the GPU produced).  Tie windows
    included: the representatives are the reference's own picks."""
    ctx, pipe, torch = env
    from eventcalib_amd import capi
    n = 240_000
    buf = SS.make_stream(n, device="cpu", seed=29)
    rec = buf.numpy()
    t0, t1 = SS.tiled_windows(5.0, 5.0 + (n - 1) / 1e6)
    S = len(t0)
    ev = buf.cuda()
    pipe.set_windows(t0, t1)
    pipe.set_detect_params(5, 36, THR)
    pipe.run(ev)
    torch.cuda.synchronize()
    packed = capi.detect_pass(ctx, ev.data_ptr(), n, t0, t1, n + 4096)
    info = pipe.win_info[:S].cpu().numpy().astype(np.int64)
    seg_off = pipe.seg_off[:2 * S].cpu().numpy().astype(np.int64)
    rep, pair, xyr = pipe.rep.cpu().numpy(), pipe.cand_pair.cpu().numpy(), pipe.cand_xyr.cpu().numpy()
    paired = tied = 0
    for s in range(S):
        lo, hi = O.window_bounds(rec, t0[s], t1[s])
        pos, neg, _ep = O.event_frame(rec, int(lo), int(hi), "reference")
        ref = O.extract_candidates(pos, neg, 4.0, 2, 5, 36, THR)
        assert info[s, 3] == ref["status"], s                      # (no ECAL_WIN_TIE_FALLBACK bit either)
        assert packed[s, 0] == ref["status"] and packed[s, 2] == len(pos) + len(neg), s
        if ref["status"]:
            assert info[s, 0] == 0 and packed[s, 1] == 0
            continue
        paired += 1
        tied += int(bool(ref["tie"]))
        op, on, m = seg_off[2 * s], seg_off[2 * s + 1], ref["n"]
        assert np.array_equal(rep[op:op + ref["nk_pos"]], ref["rep_pos"]) and np.array_equal(rep[on:on + ref["nk_neg"]], ref["rep_neg"]), s
        assert info[s, 0] == m and np.array_equal(pair[op:op + m], ref["pair"]) and np.array_equal(xyr[op:op + m], ref["xyr"]), s
        if packed[s, 1]:       # the ordered circles ecal_detect_pass hands the policy are candidates of the oracle's list
            feat = packed[s, 3:].reshape(-1, 3)
            assert all((ref["xyr"] == f).all(1).any() for f in feat), s
    assert paired >= S // 2 and tied >= 3


def test_packed_points_give_the_same_results_bit_for_bit(env):
    """ecal_packed_points (the *_packed_dev stage entry points, DetectPipeline's default): integer-pixel windows travel between
    the stages as 4-byte words, the doubles are written on request — every output array equals the plain (doubles) form, on a
    stream that mixes the cases: ordinary windows, windows of 2 000 - 9 000 events (second passes, general tiers: their
    doubles are written before those tiers read them), and windows whose coordinates are not pixels (never packed)."""
    ctx, _pipe, torch = env
    from eventcalib_amd.pipeline import DetectPipeline
    n = 400_000
    buf = SS.make_stream(n, rate=1.0e6, device="cpu", seed=31)
    t, xy, pol = SS.unpack_records(buf)
    xy = xy.clone()
    xy[150_000:170_000] += 0.25                     # 20 ms of half-way coordinates: the general slicing tiers, doubles only
    buf = SS.pack_records(t, xy, pol)
    t_first = float(t[0])
    edges = [t_first]
    for length in [1.5e-3] * 60 + [4e-3, 9e-3, 2.5e-3] + [1.5e-3] * 40 + [6e-3] + [1.5e-3] * 120:
        edges.append(edges[-1] + length)
    t0 = np.array(edges[:-1])
    t1 = np.nextafter(np.array(edges[1:]), -np.inf)
    ev = buf.cuda()
    outs = []
    for packed in (True, False):
        pipe = DetectPipeline(ctx, packed=packed)
        pipe.set_windows(t0, t1)
        pipe.set_detect_params(5, 36, THR)
        pipe.run(ev)
        torch.cuda.synchronize()
        S = len(t0)
        fmt = pipe.seg_fmt[:2 * S].cpu().numpy().copy() if packed else None
        off, cnt = pipe.seg_off[:2 * S].cpu().numpy().astype(np.int64), pipe.seg_cnt[:2 * S].cpu().numpy().astype(np.int64)
        xyo = pipe.xy.cpu().numpy()          # (packed: unpacks what has no doubles yet)
        pts = np.concatenate([xyo[off[s]:off[s] + cnt[s]] for s in range(2 * S)])
        outs.append(dict(off=off, cnt=cnt, pts=pts, ep=pipe.event_point[:n].cpu().numpy().copy(), lab=pipe.labels[:n].cpu().numpy().copy(),
                         ncl=pipe.n_clusters[:2 * S].cpu().numpy().copy(), info=pipe.win_info[:S].cpu().numpy().copy(),
                         kept=pipe.kept_labels[:n].cpu().numpy().copy(), rep=pipe.rep[:n].cpu().numpy().copy(),
                         pair=pipe.cand_pair[:n].cpu().numpy().copy(), xyr=pipe.cand_xyr[:n].cpu().numpy().copy(), fmt=fmt))
    a, b = outs
    assert (a["fmt"] == 0).sum() >= 20 and ((a["fmt"] & 1) == 1).sum() >= 300          # both forms occur ...
    big = a["cnt"][0::2] + a["cnt"][1::2]
    assert (big > 2816).any() and (big > 1408).sum() >= 3                               # ... and the later tiers are exercised
    slots = np.concatenate([np.arange(a["off"][s], a["off"][s] + a["cnt"][s]) for s in range(len(a["off"]))])
    for k in ("off", "cnt", "pts", "ep", "ncl", "info"):
        assert np.array_equal(a[k], b[k]), k
    for k in ("lab", "kept"):
        assert np.array_equal(a[k][slots], b[k][slots]), k
    for s in range(len(t0)):
        o, m = a["off"][2 * s], a["info"][s, 0]
        assert np.array_equal(a["pair"][o:o + m], b["pair"][o:o + m]) and np.array_equal(a["xyr"][o:o + m], b["xyr"][o:o + m]), s
        if a["info"][s, 3] == 0:
            for h in (0, 1):
                oo = a["off"][2 * s + h]
                assert np.array_equal(a["rep"][oo:oo + a["info"][s, 1 + h]], b["rep"][oo:oo + b["info"][s, 1 + h]]), s


def _tie_list_count(ctx, torch):
    import ctypes
    L = ctx._L
    L.ecal_debug_tie_list_count.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32), ctypes.c_void_p]
    L.ecal_debug_tie_list_count.restype = ctypes.c_int
    out = ctypes.c_uint32(0xFFFFFFFF)
    ctx._check(L.ecal_debug_tie_list_count(ctx._h, ctypes.byref(out), torch.cuda.current_stream().cuda_stream))
    return out.value


@pytest.mark.parametrize("rate", [1.0e6, 2.0e6])
def test_ties_resolved_inside_the_first_pass_equal_the_listed_form(env, rate, monkeypatch):
    """The exact extraction's first pass resolves the ties of small clusters itself (resolve_ties_inline: the DBSCAN kernel's
    kd-trees, a wave per tied cluster) instead of leaving the window on a list for the member-order launches and a second
    extraction.  Same picks either way — every output array equal to the listed form's (ECAL_FORCE=extract_no_inline_ties) and to the
    oracle's — and the list it leaves is (nearly) empty where the listed form's holds every tied window.  2 Mev/s: windows of
    ~3000 events — the second extraction pass, which resolves its ties the same way on the trees the DBSCAN kernel's second
    pass exports (segments of up to 1408 points)."""
    ctx, _pipe, torch = env
    from eventcalib_amd.capi import sync_env
    from eventcalib_amd.pipeline import DetectPipeline
    n = 300_000
    buf = SS.make_stream(n, rate=rate, device="cpu", seed=77)
    t0, t1 = SS.tiled_windows(5.0, 5.0 + (n - 1) / rate)
    ev = buf.cuda()
    S = len(t0)
    outs, counts = [], []
    ctx.set_tail_mode("tiered")      # (every size tier launched: the lean form's one tail launch per stage exports no trees)
    for listed in (False, True):
        if listed:
            monkeypatch.setenv("ECAL_FORCE", "extract_no_inline_ties")
        else:
            monkeypatch.delenv("ECAL_FORCE", raising=False)
        sync_env()
        pipe = DetectPipeline(ctx)
        pipe.set_windows(t0, t1)
        pipe.set_detect_params(5, 36, THR)
        pipe.run(ev)
        torch.cuda.synchronize()
        counts.append(_tie_list_count(ctx, torch))
        outs.append({k: getattr(pipe, k)[:n].cpu().numpy().copy() for k in ("kept_labels", "rep", "cand_pair", "cand_xyr")})
        outs[-1]["info"] = pipe.win_info[:S].cpu().numpy().copy()
        if not listed:
            exact, tied = _check_windows(pipe, torch, buf.numpy(), t0, t1, 5, 36, THR)      # == the oracle, window by window
    monkeypatch.delenv("ECAL_FORCE", raising=False)
    sync_env()
    ctx.set_tail_mode("auto")
    a, b = outs
    off = pipe.seg_off[:2 * S].cpu().numpy().astype(np.int64)
    assert np.array_equal(a["info"], b["info"])
    assert (a["info"][:, 3] & 0x100 == 0).all()                 # no window fell back to the smaller-pid pick
    for s in range(S):
        o, m = off[2 * s], a["info"][s, 0]
        assert np.array_equal(a["cand_pair"][o:o + m], b["cand_pair"][o:o + m]) and np.array_equal(a["cand_xyr"][o:o + m], b["cand_xyr"][o:o + m]), s
        if a["info"][s, 3] == 0:
            for h in (0, 1):
                oo = off[2 * s + h]
                assert np.array_equal(a["rep"][oo:oo + a["info"][s, 1 + h]], b["rep"][oo:oo + b["info"][s, 1 + h]]), s
    assert tied >= (10 if rate == 1.0e6 else 1) and counts[1] >= tied   # the listed form lists every tied window ...
    if rate == 1.0e6:
        assert counts[0] <= counts[1] // 10, counts             # ... the first pass leaves (nearly) none of them
    else:
        assert counts[0] <= counts[1] // 2, counts              # (second pass: segments of up to 1408 points, trees from the DBSCAN kernel's second pass)


@pytest.mark.parametrize("eps", [4.0, 3.0, 4.5])
def test_tie_paths_by_cluster_size_against_the_oracle(env, eps, monkeypatch):
    """The tie resolution inside the extraction pass by the size of the tied cluster: <= 16 members (hits in one register, queue
    on scalars), 17 - 64 (lists in LDS, a lane per hit), beyond 64 (the window goes down the listed path) — windows of dense
    blobs of 6 - 100 pixels centred on the diagonal x = y (a pixel and its mirror image share their norm: most medians are tied), at the shipped radius, a smaller
    integral one and a non-integral one (no pruning quirk, other disc).  Representatives, pairs and circles == the oracle's
    (the reference's nth_element over its BFS member order), == the listed form's."""
    ctx, _pipe, torch = env
    from eventcalib_amd.capi import sync_env
    from eventcalib_amd.pipeline import DetectPipeline
    rng = np.random.default_rng(int(eps * 10))
    S, win = 48, 1.5e-3
    ts, xs, ps = [], [], []
    sizes = []
    for s in range(S):
        k = 0
        pts = []
        for pol in (0, 1):
            n_blobs = int(rng.integers(4, 9))
            for b in range(n_blobs):
                # blob centres ON THE DIAGONAL x = y: a pixel (a, b) of a blob and its mirror (b, a) have the same norm, so the
                # median rank of most clusters has an equal-norm rival; the two polarities' blobs alternate along it (they pair)
                cx = cy = 20 + 30 * b + (14 if pol else 0)
                m = int(rng.choice([6, 9, 12, 16, 20, 30, 45, 64, 70, 100]))
                rad = 1.2 + 0.55 * np.sqrt(m)
                cand = [(cx + dx, cy + dy) for dx in range(-12, 13) for dy in range(-12, 13) if dx * dx + dy * dy <= rad * rad]
                rng.shuffle(cand)
                for (x, y) in cand[:m]:
                    pts.append((x, y, pol))
                sizes.append(min(m, len(cand)))
        rng.shuffle(pts)
        t = 5.0 + s * win + np.sort(rng.uniform(0.0, win * 0.98, len(pts)))
        ts.append(t)
        xs.append(np.array([(p[0], p[1]) for p in pts], float))
        ps.append(np.array([p[2] for p in pts], np.uint8))
    t = torch.tensor(np.concatenate(ts))
    xy = torch.tensor(np.concatenate(xs))
    pol = torch.tensor(np.concatenate(ps))
    buf = SS.pack_records(t, xy, pol)
    t0 = 5.0 + win * np.arange(S)
    t1 = np.nextafter(5.0 + win * np.arange(1, S + 1), -np.inf)
    ev = buf.cuda()
    n = t.shape[0]
    cluster_min, need, thr = 5, 3, 40.0
    outs, counts = [], []
    ctx.set_tail_mode("tiered")
    for listed in (False, True):
        if listed:
            monkeypatch.setenv("ECAL_FORCE", "extract_no_inline_ties")
        else:
            monkeypatch.delenv("ECAL_FORCE", raising=False)
        sync_env()
        pipe = DetectPipeline(ctx)
        pipe.set_windows(t0, t1)
        pipe.set_detect_params(cluster_min, need, thr)
        pipe.run(ev, eps, 2)
        torch.cuda.synchronize()
        counts.append(_tie_list_count(ctx, torch))
        outs.append({k: getattr(pipe, k)[:n].cpu().numpy().copy() for k in ("kept_labels", "rep", "cand_pair", "cand_xyr")})
        outs[-1]["info"] = pipe.win_info[:S].cpu().numpy().copy()
        if not listed:
            info = outs[-1]["info"].astype(np.int64)
            off = pipe.seg_off[:2 * S].cpu().numpy().astype(np.int64)
            cnt = pipe.seg_cnt[:2 * S].cpu().numpy().astype(np.int64)
            pxy = pipe.xy.cpu().numpy()
            tied = big_tied = 0
            for s in range(S):
                op, on = off[2 * s], off[2 * s + 1]
                ref = O.extract_candidates(pxy[op:op + cnt[2 * s]], pxy[on:on + cnt[2 * s + 1]], eps, 2, cluster_min, need, thr)
                assert (info[s, 3] & 0xFF) == ref["status"] and (info[s, 3] & 0x100) == 0, s
                if ref["status"]:
                    continue
                tied += int(ref["tie"])
                assert np.array_equal(outs[-1]["rep"][op:op + ref["nk_pos"]], ref["rep_pos"]), "window %d rep +" % s
                assert np.array_equal(outs[-1]["rep"][on:on + ref["nk_neg"]], ref["rep_neg"]), "window %d rep -" % s
                assert info[s, 0] == ref["n"] and np.array_equal(outs[-1]["cand_pair"][op:op + ref["n"]], ref["pair"]), s
                assert np.array_equal(outs[-1]["cand_xyr"][op:op + ref["n"]], ref["xyr"]), s
    monkeypatch.delenv("ECAL_FORCE", raising=False)
    sync_env()
    ctx.set_tail_mode("auto")
    a, b = outs
    for k in ("info", "rep", "cand_pair", "cand_xyr"):
        if k == "info":
            assert np.array_equal(a[k], b[k])
    assert tied >= S // 2, tied                                   # most windows hold tied clusters ...
    assert max(sizes) > 64 and min(sizes) <= 16                   # ... of every size class
    assert 0 < counts[0] < counts[1], counts                      # windows with a tied cluster beyond 64 members stay on the list, the others do not
