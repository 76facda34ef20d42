"""Test helper: EVERY window of a device pipeline run against the CPU oracle's whole-stream loop
(oracle_detect_windows_full_mt: EventFrame.cpp:10-36 -> dbscan.h:115-265 -> CirclesEventFrame.cpp:89-312 per window, on
all host threads), array by array in the pipeline's own slot layout — no sampling.  The k-d tree under the oracle's
DBSCAN is the reference's compiled kdtree.cpp (oracle/_ref) whenever that build is present; the caller is told which."""
import time

import numpy as np

import oracle_lib as O


def use_reference_kdtree_if_present():
    """Select the oracle's kd backend for this process and say which one it is (printed by the tests, so the log shows what
    checked the kernels)."""
    O.set_kd_backend(O.have_ref_kdtree())
    return O.kd_backend()


def _first_bad_window(bad_slot_mask, win_base, torch):
    idx = int(torch.nonzero(bad_slot_mask)[0])
    return int(np.searchsorted(win_base, idx, side="right") - 1), idx


def compare_all_windows(pipe, rec, t0, t1, torch, eps=4.0, minpts=2, det=(5, 36, 15.511363636363637), fit_circle=False, knn_num=3,
                        n_threads=None, xyr_exact=True):
    """pipe: a DetectPipeline after run() on the packed records `rec` (numpy uint8, host copy of what the device holds) with
    windows t0 / t1.  Asserts that window bounds, point sets in the reference's order, event->point map, labels, cluster
    counts, kept labels, window verdicts, representatives, pairs and circles of ALL windows equal the oracle's, bit for bit.
    Returns counts for the caller's sanity floor (windows, paired, tied, points, candidates, seconds of oracle time)."""
    S = len(t0)
    wb = pipe.win_base[:S + 1].cpu().numpy().astype(np.uint64)
    slots = int(wb[-1])
    tic = time.time()
    f = O.detect_windows_full(rec, t0, t1, wb, slots, eps, minpts, det[0], det[1], det[2], fit_circle, knn_num, n_threads)
    oracle_s = time.time() - tic
    assert np.array_equal(pipe.win_lo[:S].cpu().numpy().astype(np.uint64), f["win_lo"]), "window lower bounds"
    assert np.array_equal(pipe.win_hi[:S].cpu().numpy().astype(np.uint64), f["win_hi"]), "window upper bounds"
    cnt = pipe.seg_cnt[:2 * S].cpu().numpy().astype(np.uint32)
    bad = np.nonzero(cnt != f["seg_cnt"])[0]
    assert bad.size == 0, "point counts differ in %d segments, first: window %d" % (bad.size, bad[0] // 2)
    off = pipe.seg_off[:2 * S].cpu().numpy().astype(np.uint64)
    assert np.array_equal(off[0::2], wb[:-1]) and np.array_equal(off[1::2], wb[:-1] + cnt[0::2])
    ncl = pipe.n_clusters[:2 * S].cpu().numpy().astype(np.uint32)
    bad = np.nonzero(ncl != f["n_clusters"])[0]
    assert bad.size == 0, "cluster counts differ in %d segments, first: window %d" % (bad.size, bad[0] // 2)
    info = pipe.win_info[:S].cpu().numpy().astype(np.uint32)
    assert not (info[:, 3] & 0x100).any(), "a window carries ECAL_WIN_TIE_FALLBACK"
    bad = np.nonzero((info != f["win_info"]).any(axis=1))[0]
    assert bad.size == 0, "window verdicts differ in %d windows, first: %d (%s vs %s)" % (
        bad.size, bad[0], info[bad[0]] if bad.size else "", f["win_info"][bad[0]] if bad.size else "")
    dev = pipe.labels.device

    def mask(name):
        return torch.from_numpy(f[name]).to(dev).bool()

    def check(what, got, want_np, m):
        want = torch.from_numpy(want_np.view(np.int32) if want_np.dtype == np.uint32 else want_np).to(dev)
        got = got[:slots]
        if want.dtype == torch.float64:         # bit patterns, not values: -0.0 / NaN must not pass as equal / unequal
            got, want = got.contiguous().view(torch.int64), want.view(torch.int64)
        elif got.dtype != want.dtype:
            got = got.to(want.dtype)
        ne = got != want
        if ne.dim() > 1:
            ne = ne.any(dim=1)
        ne &= m
        nb = int(ne.sum())
        if nb:
            w, idx = _first_bad_window(ne, wb, torch)
            raise AssertionError("%s differs in %d slots, first: slot %d of window %d" % (what, nb, idx, w))

    m_pts = mask("def_pts")
    check("event -> point map", pipe.event_point, f["event_point"], torch.ones(slots, dtype=torch.bool, device=dev))
    check("points (reference order)", pipe.xy, f["xy"], m_pts)
    check("DBSCAN labels", pipe.labels, f["labels"], m_pts)
    check("kept labels", pipe.kept_labels, f["kept_labels"], mask("def_kept"))
    check("representatives", pipe.rep, f["rep"], mask("def_rep"))
    m_c = mask("def_cand")
    check("candidate pairs", pipe.cand_pair, f["cand_pair"], m_c)
    if xyr_exact:
        check("candidate circles", pipe.cand_xyr, f["cand_xyr"], m_c)
    else:       # fitCircle == 1: the fit's sums run in another order (documented tolerance 1e-9)
        d = (pipe.cand_xyr[:slots] - torch.from_numpy(f["cand_xyr"]).to(dev)).abs().amax(dim=1)
        assert float(d[m_c].max() if int(m_c.sum()) else 0.0) < 1e-9
    ok = f["win_info"][:, 3] == 0
    return dict(windows=S, paired=int(ok.sum()), tied=int((f["tie"] != 0).sum()), points=int(f["def_pts"].sum()),
                candidates=int(f["def_cand"].sum()), events=f["events"], oracle_seconds=oracle_s, threads=f["n_threads"],
                kd_backend=O.kd_backend())
