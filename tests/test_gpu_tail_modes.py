"""GPU: the stage calls' tail scheduling (ecal_set_tail_mode, include/ecal.h).  A stage's first-pass kernel lists what it cannot
take; "tiered" launches every later size tier behind it, "lean" ONE general tail launch (slicing: the global-scratch tier;
DBSCAN: dbscan_tail_kernel; member order: the 4096-point launch over both lists + the global-scratch launch), "auto" picks
lean while the stage's lists were empty at its previous call.  The choice must move time only: every output array identical,
bit for bit, in all three modes — on the shipped configuration (lists empty) and on streams that fill the lists (windows from
empty to 20 k events, non-pixel coordinates, other radii), and equal to the CPU oracle."""
import numpy as np
import pytest

import synth_stream as SS
import test_gpu_fused as TF

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    import eventcalib_amd
    from eventcalib_amd.pipeline import DetectPipeline
    ctx = eventcalib_amd.Context(0)
    yield ctx, DetectPipeline, torch
    ctx.close()


def _all_modes(env, ev, t0, t1, eps=4.0, minpts=2, det=None, slots=None, exact=True, passes=2):
    ctx, DetectPipeline, torch = env
    S = len(t0)
    snaps = {}
    try:
        for mode in ("tiered", "lean", "auto"):
            ctx.set_tail_mode(mode)
            pipe = DetectPipeline(ctx)
            pipe.set_windows(t0, t1)
            if det:
                pipe.set_detect_params(*det)
            for _ in range(passes):          # auto: the second pass is scheduled from what the first one saw
                pipe._ensure(S, slots if slots else ev.numel() // 25)
                TF._poison(pipe)
                pipe.run(ev, eps, minpts, slots=slots, exact_ties=exact)
                torch.cuda.synchronize()
            assert not pipe.overflowed()
            snaps[mode] = (TF._snapshot(pipe, S, torch), pipe)
    finally:
        ctx.set_tail_mode("auto")
    ref = snaps["tiered"][0]
    for mode in ("lean", "auto"):
        got = snaps[mode][0]
        for k in ref:
            assert ref[k].shape == got[k].shape, (mode, k)
            if ref[k].dtype.is_floating_point:
                assert torch.equal(ref[k].view(torch.int64), got[k].view(torch.int64)), (mode, k)
            else:
                assert torch.equal(ref[k], got[k]), (mode, k)
    return ref, snaps["lean"][1]


def test_shipped_configuration(env):
    ctx, _, torch = env
    n = 2_000_000
    ev = SS.make_stream(n, device="cuda", seed=4)
    t0, t1 = SS.tiled_windows(5.0, 5.0 + (n - 1) / 1e6)
    ref, pipe = _all_modes(env, ev, t0, t1, det=(5, 36, 15.511363636363637))
    assert int((ref["win_info"][:, 3] == 0).sum()) > 600
    import full_compare as FC
    st = FC.compare_all_windows(pipe, ev.cpu().numpy(), t0, t1, torch)          # the lean run == the oracle, every window
    assert st["paired"] > 600 and st["tied"] > 50


def test_windows_of_every_size_fill_the_lists(env):
    """Window lengths from empty to 20 k events over a 2 Mev/s stream: every slicing tier, DBSCAN segments beyond the pixel
    kernel's first and second pass and beyond 4096 points, member order beyond its first launch, extraction beyond its LDS staging —
    in lean mode all of that goes through the general tails; and the whole result == the oracle."""
    ctx, _, torch = env
    n = 1_500_000
    ev = SS.make_stream(n, rate=2.0e6, device="cuda", seed=9, noise_frac=0.5)      # (noise: enough distinct pixels for > 4096-point segments)
    rng = np.random.default_rng(3)
    lens = rng.choice([0.0, 2e-6, 2e-5, 4e-4, 9e-4, 1.3e-3, 1.9e-3, 3e-3, 6e-3, 1e-2, 2e-2], size=300)
    starts = 5.0 + rng.uniform(0, n / 2.0e6 - 2e-2, size=300)
    t0, t1 = starts, starts + lens            # overlapping, unordered windows
    t1[7] = t0[7] - 1e-3                      # an empty window (end before start)
    ref, pipe = _all_modes(env, ev, t0, t1, slots=8_000_000, det=(5, 36, 15.511363636363637))
    cnt = ref["seg_cnt"]
    assert int(cnt.max()) > 4096 and int((cnt == 0).sum()) > 0 and int(((cnt > 0) & (cnt < 20)).sum()) > 0, int(cnt.max())
    import full_compare as FC
    st = FC.compare_all_windows(pipe, ev.cpu().numpy(), t0, t1, torch)
    assert st["paired"] > 5, st


@pytest.mark.parametrize("eps,minpts", [(3.0, 3), (4.5, 2), (17.0, 2)])
def test_other_radii(env, eps, minpts):
    ctx, _, torch = env
    n = 500_000
    ev = SS.make_stream(n, rate=1.6e6, device="cuda", seed=21)
    t0, t1 = SS.tiled_windows(5.0, 5.0 + (n - 1) / 1.6e6)
    _all_modes(env, ev, t0, t1, eps=eps, minpts=minpts, passes=1)


def test_non_pixel_coordinates(env):
    ctx, _, torch = env
    n = 400_000
    ev = SS.make_stream(n, device="cuda", seed=33).clone()
    rec = ev.view(-1, 25)
    xs = rec[:, 8:16].contiguous().view(torch.float64).view(-1)
    idx = torch.arange(n, device="cuda")
    sel = (idx // 1500) % 9 == 4
    xs[sel] = xs[sel] + 0.5          # half-pixel x in the events of a few windows: general tiers in every stage
    rec[:, 8:16] = xs.view(-1, 1).view(torch.uint8)
    t0, t1 = SS.tiled_windows(5.0, 5.0 + (n - 1) / 1e6)
    ref, pipe = _all_modes(env, ev, t0, t1, det=(5, 36, 15.511363636363637))
    import full_compare as FC
    FC.compare_all_windows(pipe, ev.cpu().numpy(), t0, t1, torch)


def test_auto_goes_lean_after_an_empty_pass_and_back(env):
    """The feedback itself: after a pass whose lists were empty the next call is scheduled lean, after a pass that listed
    work every tier is launched again — observed through the kernel count of a pass (rocprof-free: the context's own counter
    of launches is not exposed, so this checks results only across the switch: tiled pass, big-window pass, tiled pass)."""
    ctx, DetectPipeline, torch = env
    n = 600_000
    ev = SS.make_stream(n, rate=2.0e6, device="cuda", seed=12)
    a0, a1 = SS.tiled_windows(5.0, 5.0 + (n - 1) / 2.0e6, 7.5e-4)        # ~1500-event windows: first-pass work
    b0 = 5.0 + 0.02 * np.arange(10)
    b1 = b0 + 6e-3                                                       # 12 k-event windows: tails
    outs = []
    ctx.set_tail_mode("auto")
    for (t0, t1) in ((a0, a1), (a0, a1), (b0, b1), (a0, a1), (b0, b1), (b0, b1), (a0, a1)):
        pipe = DetectPipeline(ctx)
        pipe.set_windows(t0, t1)
        pipe.set_detect_params(5, 36, 15.511363636363637)
        pipe.run(ev, slots=n)
        torch.cuda.synchronize()
        outs.append(TF._snapshot(pipe, len(t0), torch))
    for i, j in ((0, 1), (0, 3), (0, 6), (2, 4), (2, 5)):
        for k in outs[i]:
            a, b = outs[i][k], outs[j][k]
            assert torch.equal(a.view(torch.int64) if a.dtype.is_floating_point else a, b.view(torch.int64) if b.dtype.is_floating_point else b), (i, j, k)


def test_auto_drops_the_tiers_behind_the_second_pass_and_takes_them_back(env):
    """The in-between plan (ecal_tail_plan: SEMI): after a pass whose FIRST-pass lists held work and whose second-pass lists were
    empty (windows of 2 - 4 k events: the keyframe search's kind), slicing and DBSCAN run first pass + second pass + ONE general
    launch.  The pass after that is scheduled from stale news: windows of 12 k events and windows with half-pixel coordinates then
    go through that one launch — every array must equal the all-tiers form's, bit for bit."""
    ctx, DetectPipeline, torch = env
    n = 600_000
    ev = SS.make_stream(n, rate=2.0e6, device="cuda", seed=12)
    evh = ev.clone()
    rec = evh.view(-1, 25)
    xs = rec[:, 8:16].contiguous().view(torch.float64).view(-1)
    sel = (torch.arange(n, device="cuda") // 3000) % 5 == 2
    xs[sel] = xs[sel] + 0.5
    rec[:, 8:16] = xs.view(-1, 1).view(torch.uint8)
    c0, c1 = SS.tiled_windows(5.0, 5.0 + (n - 1) / 2.0e6, 1.5e-3)        # ~3000-event windows: second-pass work, nothing behind it
    b0 = 5.0 + 0.02 * np.arange(10)
    b1 = b0 + 6e-3                                                       # 12 k-event windows: work behind the second pass
    seq = ((ev, c0, c1), (ev, c0, c1), (ev, b0, b1), (ev, c0, c1), (ev, c0, c1), (evh, c0, c1), (evh, c0, c1), (ev, c0, c1), (evh, b0, b1))

    def run_all(mode):
        ctx.set_tail_mode(mode)
        outs = []
        for (e, t0, t1) in seq:
            pipe = DetectPipeline(ctx)
            pipe.set_windows(t0, t1)
            pipe.set_detect_params(5, 36, 15.511363636363637)
            pipe.run(e, slots=n)
            torch.cuda.synchronize()
            assert not pipe.overflowed()
            outs.append(TF._snapshot(pipe, len(t0), torch))
        return outs
    try:
        tiered = run_all("tiered")
        auto = run_all("auto")
    finally:
        ctx.set_tail_mode("auto")
    assert int(tiered[0]["seg_cnt"].max()) > 768 and int(tiered[2]["seg_cnt"].max()) > 2048
    for i, (a, b) in enumerate(zip(tiered, auto)):
        for k in a:
            x, y = a[k], b[k]
            assert torch.equal(x.view(torch.int64) if x.dtype.is_floating_point else x, y.view(torch.int64) if y.dtype.is_floating_point else y), (i, k)
