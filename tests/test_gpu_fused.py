"""GPU: ecal_detect_fused_dev (one kernel carries a window through slicing, both DBSCAN runs and candidate extraction)
against the three stage entry points it stands for (ecal_slice_events_dev + ecal_dbscan_batch_dev + ecal_extract_batch_dev,
each of which has its own oracle tests): every output array identical, bit for bit — including the windows the fused kernel
hands to the stages' later passes (too many events, too many points, empty, non-pixel coordinates)."""
import numpy as np
import pytest

import synth_stream as SS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    import eventcalib_amd
    from eventcalib_amd.pipeline import DetectPipeline
    ctx = eventcalib_amd.Context(0)
    yield ctx, DetectPipeline, torch
    ctx.close()


def _snapshot(pipe, S, torch):
    """Everything the pass defines, cut to the defined slots (the arrays' other slots are never written)."""
    off, cnt = pipe.seg_off[:2 * S].long(), pipe.seg_cnt[:2 * S].long()
    info = pipe.win_info[:S].long()
    dev = off.device
    total = int(cnt.sum())
    seg_of = torch.repeat_interleave(torch.arange(2 * S, device=dev), cnt)
    slot = torch.repeat_interleave(off, cnt) + (torch.arange(total, device=dev) - torch.repeat_interleave(torch.cumsum(cnt, 0) - cnt, cnt))
    n = min(int(pipe.win_base[S]), pipe.event_point.numel())     # events of all windows = slots in use
    out = {
        "seg_off": off, "seg_cnt": cnt, "n_clusters": pipe.n_clusters[:2 * S].long(), "win_info": info,
        "event_point": pipe.event_point[:n].clone(), "xy": pipe.xy[slot].clone(), "labels": pipe.labels[slot].clone(),
        "kept_labels": pipe.kept_labels[slot].clone(), "overflow": pipe.flags[:1].clone(),
    }
    # candidates: the first win_info[s, 0] entries from the + segment's first slot
    nc = info[:, 0]
    cslot = torch.repeat_interleave(off[0::2], nc) + (torch.arange(int(nc.sum()), device=dev) -
                                                      torch.repeat_interleave(torch.cumsum(nc, 0) - nc, nc))
    out["cand_pair"] = pipe.cand_pair[cslot].clone()
    out["cand_xyr"] = pipe.cand_xyr[cslot].clone()
    # representatives: of the kept clusters of both polarities, in the windows that reach the pairing
    ok = info[:, 3] == 0
    nk = torch.stack([info[:, 1], info[:, 2]], 1).reshape(-1) * ok.repeat_interleave(2)
    rslot = torch.repeat_interleave(off, nk) + (torch.arange(int(nk.sum()), device=dev) - torch.repeat_interleave(torch.cumsum(nk, 0) - nk, nk))
    out["rep"] = pipe.rep[rslot].clone()
    return out


def _poison(pipe):
    for name in ("xy", "event_point", "labels", "kept_labels", "rep", "cand_pair", "cand_xyr", "seg_off", "seg_cnt", "n_clusters", "win_info"):
        t = getattr(pipe, name)
        t.view(-1).view(dtype=__import__("torch").uint8).fill_(0x5A)


def _both(env, ev, t0, t1, eps=4.0, minpts=2, hints=(0, 0), det=None, slots=None):
    ctx, DetectPipeline, torch = env
    pipe = DetectPipeline(ctx)
    pipe.set_windows(t0, t1)
    if det:
        pipe.set_detect_params(*det)
    S = len(t0)
    pipe.run(ev, eps, minpts, slots=slots, max_win_events=hints[0], max_seg_points=hints[1], exact_ties=False)   # the three primitives
    torch.cuda.synchronize()
    assert not pipe.overflowed()
    ref = _snapshot(pipe, S, torch)
    _poison(pipe)
    pipe.run(ev, eps, minpts, slots=slots, max_win_events=hints[0], max_seg_points=hints[1], fused=True)
    torch.cuda.synchronize()
    got = _snapshot(pipe, S, torch)
    for k in ref:
        assert ref[k].shape == got[k].shape, k
        if ref[k].dtype.is_floating_point:   # bit for bit (NaN-safe)
            assert torch.equal(ref[k].view(torch.int64), got[k].view(torch.int64)), k
        else:
            assert torch.equal(ref[k], got[k]), k
    return ref


def test_shipped_configuration_every_array_identical(env):
    ctx, _, torch = env
    n = 3_000_000
    ev = SS.make_stream(n, device="cuda", seed=4)
    t0, t1 = SS.tiled_windows(5.0, 5.0 + (n - 1) / 1e6)
    ref = _both(env, ev, t0, t1)
    assert int((ref["win_info"][:, 3] == 0).sum()) > 1000      # the pass does reach the pairing
    assert int(ref["win_info"][:, 0].sum()) > 36000


def test_windows_the_fused_kernel_hands_on(env):
    """Window lengths from empty to 20 k events over a 2 Mev/s stream: first pass, second pass (<= 4095 events), general tiers;
    segments beyond the pixel DBSCAN's first (1024 points) and second pass (2048); extraction windows beyond its LDS staging."""
    ctx, _, torch = env
    n = 1_500_000
    ev = SS.make_stream(n, rate=2.0e6, device="cuda", seed=9)
    rng = np.random.default_rng(3)
    lens = rng.choice([0.0, 2e-5, 4e-4, 9e-4, 1.3e-3, 1.9e-3, 3e-3, 6e-3, 1e-2], size=400)
    starts = 5.0 + rng.uniform(0, n / 2.0e6 - 1e-2, size=400)
    t0, t1 = starts, starts + lens            # overlapping, unordered windows
    t1[7] = t0[7] - 1e-3                      # an empty window (end before start)
    ref = _both(env, ev, t0, t1, slots=6_000_000)
    cnt = ref["seg_cnt"]
    assert int(cnt.max()) > 2048 and int((cnt == 0).sum()) > 0
    assert int((ref["win_info"][:, 3] == 0).sum()) > 20


@pytest.mark.parametrize("eps,minpts", [(4.0, 2), (3.0, 3), (4.5, 2), (17.0, 2)])
def test_other_radii(env, eps, minpts):
    """eps 4 = the compiled-in disc; other radii below 16 = the generic pixel form inside the fused kernel; eps >= 16 = no pixel
    form at all, i.e. the three stage functions one after the other."""
    ctx, _, torch = env
    n = 600_000
    ev = SS.make_stream(n, device="cuda", seed=21)
    t0, t1 = SS.tiled_windows(5.0, 5.0 + (n - 1) / 1e6)
    _both(env, ev, t0, t1, eps=eps, minpts=minpts)


def test_non_pixel_coordinates_and_fit_circle(env):
    ctx, _, torch = env
    n = 400_000
    ev = SS.make_stream(n, device="cuda", seed=33).clone()
    rec = ev.view(-1, 25)
    # half-pixel x in the events of a few windows: those windows leave the pixel slicer for the general tier
    xs = rec[:, 8:16].contiguous().view(torch.float64).view(-1)
    idx = torch.arange(n, device="cuda")
    sel = (idx // 1500) % 9 == 4
    xs[sel] = xs[sel] + 0.5
    rec[:, 8:16] = xs.view(-1, 1).view(torch.uint8)
    t0, t1 = SS.tiled_windows(5.0, 5.0 + (n - 1) / 1e6)
    _both(env, ev, t0, t1)
    # fitCircle == 1 (the k-nearest / algebraic-fit pairing): not compiled into the fused kernel -> three stage calls
    _both(env, ev, t0, t1, det=(5, 36, 15.511363636363637, True, 3))


def test_first_occurrence_order_takes_the_stage_functions(env):
    import eventcalib_amd.capi as capi
    ctx, _, torch = env
    n = 300_000
    ev = SS.make_stream(n, device="cuda", seed=5)
    t0, t1 = SS.tiled_windows(5.0, 5.0 + (n - 1) / 1e6)
    ctx.set_point_order(capi.Context.ORDER_FIRST_OCCURRENCE)
    try:
        _both(env, ev, t0, t1)
    finally:
        ctx.set_point_order(capi.Context.ORDER_REFERENCE)


def test_capacity_overflow_is_reported_the_same(env):
    ctx, DetectPipeline, torch = env
    n = 200_000
    ev = SS.make_stream(n, device="cuda", seed=6)
    t0, t1 = SS.tiled_windows(5.0, 5.0 + (n - 1) / 1e6)
    pipe = DetectPipeline(ctx)
    pipe.set_windows(t0, t1)
    pipe._ensure(len(t0), n)
    for fused in (False, True):
        pipe.flags.zero_()
        pipe.run(ev, slots=n // 2, fused=fused)       # half the slots: the later windows do not fit
        torch.cuda.synchronize()
        assert pipe.overflowed()
        S = len(t0)
        cnt = pipe.seg_cnt[:2 * S].long()
        assert int(cnt[-2:].sum()) == 0 and int(cnt[:2].sum()) > 0
