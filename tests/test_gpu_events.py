"""GPU parity: window bounds + EventFrame slicing + DBSCAN on the sliced sets, vs the oracle."""
import numpy as np
import pytest

import oracle_lib as O
import synth_stream as SS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["reference", "first"])
def env(request):
    """Every test runs in both element orders: the reference's (EventFrame.cpp:34-35 on libstdc++, checked against the
    oracle's real std::unordered_set) and first occurrence (checked against the oracle's canonical mode)."""
    import torch
    import eventcalib_amd
    from eventcalib_amd.pipeline import DetectPipeline
    ctx = eventcalib_amd.Context(0)
    assert ctx.point_order() == "reference"          # the default is the reference's order
    ctx.set_point_order(request.param)
    yield ctx, DetectPipeline(ctx), torch
    ctx.close()


def _compare(pipe, torch, rec_np, t0, t1, eps=4.0, minpts=2, check_labels=True):
    S = len(t0)
    order = {"reference": "reference", "first": "canonical"}[pipe.ctx.point_order()]
    lo = pipe.win_lo[:S].cpu().numpy().astype(np.int64)
    hi = pipe.win_hi[:S].cpu().numpy().astype(np.int64)
    base = pipe.win_base[:S + 1].cpu().numpy().astype(np.int64)
    seg_off = pipe.seg_off[:2 * S].cpu().numpy().astype(np.int64)
    seg_cnt = pipe.seg_cnt[:2 * S].cpu().numpy().astype(np.int64)
    xy = pipe.xy.cpu().numpy()
    ep = pipe.event_point.cpu().numpy()
    labels = pipe.labels.cpu().numpy()
    ncl = pipe.n_clusters[:2 * S].cpu().numpy()
    assert not pipe.overflowed()
    run = 0
    for s in range(S):
        olo, ohi = O.window_bounds(rec_np, t0[s], t1[s])
        assert (lo[s], hi[s]) == (olo, ohi), "window %d bounds" % s
        assert base[s] == run
        run += ohi - olo
        pos, neg, oep = O.event_frame(rec_np, olo, ohi, order)
        assert seg_cnt[2 * s] == pos.shape[0] and seg_cnt[2 * s + 1] == neg.shape[0], "window %d counts" % s
        if ohi > olo:
            assert seg_off[2 * s] == base[s] and seg_off[2 * s + 1] == base[s] + pos.shape[0]
        gp = xy[seg_off[2 * s]:seg_off[2 * s] + seg_cnt[2 * s]]
        gn = xy[seg_off[2 * s + 1]:seg_off[2 * s + 1] + seg_cnt[2 * s + 1]]
        assert np.array_equal(gp, pos) and np.array_equal(gn, neg), "window %d points" % s
        assert np.array_equal(ep[base[s]:base[s] + (ohi - olo)], oep), "window %d event_point" % s
        if check_labels:
            for k, pts in ((0, pos), (1, neg)):
                if pts.shape[0] == 0:
                    assert ncl[2 * s + k] == 0
                    continue
                rc, ol, onc = O.dbscan(pts, eps, minpts)
                o = seg_off[2 * s + k]
                assert np.array_equal(labels[o:o + pts.shape[0]], ol), "window %d pol %d labels" % (s, k)
                assert ncl[2 * s + k] == onc
    assert base[S] == run


def test_tiled_windows_on_synthetic_stream(env):
    ctx, pipe, torch = env
    buf = SS.make_stream(60000, device="cpu")
    t, _, _ = SS.unpack_records(buf)
    t0, t1 = SS.tiled_windows(float(t[0]), float(t[-1]))
    pipe.set_windows(t0, t1)
    pipe.run(buf.cuda())
    torch.cuda.synchronize()
    _compare(pipe, torch, buf.numpy(), t0, t1)


def test_overlapping_empty_and_growing_windows(env):
    """The reference's adaptive policy produces overlapping windows of 3..9 steps (eventCameraCalib.cpp:49-81)."""
    ctx, pipe, torch = env
    buf = SS.make_stream(40000, device="cpu", seed=7)
    t, _, _ = SS.unpack_records(buf)
    ts, te = float(t[0]), float(t[-1])
    step = 5e-4
    t0 = [ts, ts + step, ts + 2 * step, ts - 1.0, te + 1.0, ts + 10 * step, ts + 3 * step, ts]
    t1 = [ts + 3 * step, ts + 5 * step, ts + 11 * step, ts - 0.5, te + 2.0, ts + 10 * step, ts + 3 * step + 1e-7, te]
    pipe.set_windows(t0, t1)
    pipe.run(buf.cuda(), slots=200000)
    torch.cuda.synchronize()
    _compare(pipe, torch, buf.numpy(), t0, t1)


def test_slice_size_tiers_and_duplicates(env):
    """Windows in every slicer tier (<=2048, <=5120 LDS; > 5120 global scratch), heavy duplication,
    +/- cancellation, non-integer and negative-zero coordinates."""
    ctx, pipe, torch = env
    rng = np.random.default_rng(5)
    n = 30000
    t = np.sort(rng.uniform(0, 1, n))
    x = rng.integers(0, 60, n).astype(np.float64) * rng.choice([1.0, 0.5], n)
    y = rng.integers(0, 40, n).astype(np.float64)
    x[rng.random(n) < 0.01] = -0.0
    p = (rng.random(n) < 0.5).astype(np.uint8) * rng.integers(1, 255, n).astype(np.uint8)
    rec = O.pack_events(t, x, y, p)
    q = [0.0, t[1500], t[1501], t[5000], t[5001], t[12000], t[12001], t[29999]]
    t0 = [q[0], q[2], q[4], q[6], 0.0]
    t1 = [q[1], q[3], q[5], q[7], 1.0]
    pipe.set_windows(t0, t1)
    pipe.run(torch.from_numpy(rec).cuda(), slots=80000)
    torch.cuda.synchronize()
    _compare(pipe, torch, rec, t0, t1, check_labels=True)


def test_overflow_is_reported(env):
    ctx, pipe, torch = env
    buf = SS.make_stream(5000, device="cpu")
    t, _, _ = SS.unpack_records(buf)
    pipe.set_windows([float(t[0])], [float(t[-1])])
    pipe._cap_slots = 0
    pipe.run(buf.cuda(), slots=100)
    torch.cuda.synchronize()
    assert pipe.overflowed()
    assert int(pipe.seg_cnt[0]) == 0 and int(pipe.seg_cnt[1]) == 0


def test_check_sorted(env):
    ctx, pipe, torch = env
    buf = SS.make_stream(5000, device="cpu").cuda()
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    ctx.check_sorted_dev(buf.data_ptr(), 5000, flag.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert int(flag.item()) == 0
    rec = buf.reshape(5000, 25).flip(0).contiguous().reshape(-1)
    ctx.check_sorted_dev(rec.data_ptr(), 5000, flag.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert int(flag.item()) == 1


@pytest.mark.parametrize("shift", [0.0, 25.0])
def test_pixel_and_general_slicers_agree(env, shift):
    """Sensor-pixel windows take the hash slicer (its second pass above 2047 events); ECAL_FORCE=slice_general sends everything through
    the general tiers.  Same outputs, and both equal the oracle.  shift = 0: negative coordinates and -0.0 (the hash kernel hands
    those windows to the general tiers); shift = 25: all coordinates >= 0.  Both: both-polarity cancellation, a window with one
    non-integer coordinate, a window above the first pass's capacity."""
    import os
    ctx, pipe, torch = env
    rng = np.random.default_rng(15)
    n = 12000
    t = np.sort(rng.uniform(0, 1, n))
    x = rng.integers(-20, 60, n).astype(np.float64) + shift
    y = rng.integers(-10, 40, n).astype(np.float64) + shift
    if shift == 0.0:
        x[rng.random(n) < 0.01] = -0.0
    x[7000] = 12.5                                         # one fractional coordinate -> that window is not "pixels"
    p = (rng.random(n) < 0.5).astype(np.uint8)
    rec = O.pack_events(t, x, y, p)
    cuts = [0, 1500, 3100, 5000, 6900, 8400, 12000]        # 1500, 1600, 1900, 1900, 1500 (fractional), 3600 (> 2047)
    t0 = [t[a] for a in cuts[:-1]]
    t1 = [t[b - 1] for b in cuts[1:]]
    d = torch.from_numpy(rec).cuda()
    outs = []
    for var in (None, "ECAL_FORCE"):
        if var:
            os.environ[var] = "slice_general"
            __import__("eventcalib_amd.capi", fromlist=["sync_env"]).sync_env()   # (the switches are read once per context)
        try:
            pipe.set_windows(t0, t1)
            pipe.run(d, slots=20000)
            torch.cuda.synchronize()
            _compare(pipe, torch, rec, t0, t1, check_labels=False)
            outs.append((pipe.xy[:12000].cpu().numpy().copy(), pipe.event_point[:12000].cpu().numpy().copy(),
                         pipe.seg_cnt[:12].cpu().numpy().copy()))
        finally:
            if var:
                os.environ.pop(var, None)
                __import__("eventcalib_amd.capi", fromlist=["sync_env"]).sync_env()   # (the switches are read once per context)
    n_pts = int(outs[0][2].sum())
    for o in outs[1:]:
        assert np.array_equal(outs[0][2], o[2]) and np.array_equal(outs[0][1], o[1])
        assert np.array_equal(outs[0][0][:n_pts].view(np.int64), o[0][:n_pts].view(np.int64))


def test_window_bounds_search_edge_cases(env):
    """ecal_window_bounds_dev interpolates the start of its searches (time grows ~linearly with the index), gallops to a
    bracket and bisects: against numpy's searchsorted on streams whose rate is far from constant (bursts, long pauses,
    runs of equal time stamps), for windows before / after / across the stream, inverted, infinite and NaN."""
    import torch
    ctx = env[0]
    rng = np.random.default_rng(3)
    for n in (1, 2, 5, 1000, 200_000):
        gaps = rng.exponential(1.0, n) * rng.choice([1e-6, 1e-6, 1e-3, 0.0, 5.0], n, p=[0.5, 0.2, 0.1, 0.19, 0.01])
        t = 7.0 + np.cumsum(gaps)
        rec = np.zeros((n, 25), np.uint8)
        rec[:, 0:8] = t.view(np.uint8).reshape(n, 8)
        ev = torch.from_numpy(rec.reshape(-1)).cuda()
        S, shift = (4000, 0) if n != 1000 else (4099, 1)   # (shift: outputs not 16-byte aligned — the scan's narrow path)
        a = rng.choice(t, S) + rng.choice([0.0, 0.0, 1e-9, -1e-9, 1e-3, -1e-3], S)
        b = a + rng.choice([0.0, 1e-6, 1e-3, 1.0, -1e-3], S)
        a[:8] = [-np.inf, np.inf, t[0] - 1, t[-1] + 1, t[0], t[-1], np.nan, -np.inf]
        b[:8] = [np.inf, -np.inf, t[0] - 0.5, t[-1] + 2, t[0], t[-1], t[-1], np.nan]
        d_a, d_b = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        lo = torch.zeros(S + shift, dtype=torch.int32, device="cuda")[shift:]
        hi = torch.zeros(S + shift, dtype=torch.int32, device="cuda")[shift:]
        base = torch.zeros(S + 1 + shift, dtype=torch.int32, device="cuda")[shift:]
        ctx.window_bounds_dev(ev.data_ptr(), n, d_a.data_ptr(), d_b.data_ptr(), S, lo.data_ptr(), hi.data_ptr(), base.data_ptr(),
                              torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        ok = ~(np.isnan(a) | np.isnan(b))
        want_lo = np.searchsorted(t, a, side="left")       # std::lower_bound(t0), EventFrame.cpp:14
        want_hi = np.maximum(np.searchsorted(t, b, side="right"), want_lo)   # std::upper_bound(t1), :15
        assert np.array_equal(lo.cpu().numpy()[ok], want_lo[ok]), n
        assert np.array_equal(hi.cpu().numpy()[ok], want_hi[ok]), n
        # NaN bounds: every comparison is false -> both searches end at 0, the window is empty (as the bisection they replace)
        got_lo, got_hi = lo.cpu().numpy(), hi.cpu().numpy()
        assert got_lo[6] == 0 and got_hi[7] == got_lo[7]
        assert np.array_equal(base.cpu().numpy(), np.concatenate([[0], np.cumsum(got_hi.astype(np.int64) - got_lo)]).astype(np.int32))


def test_single_odd_event_sends_the_window_to_the_general_slicer(env):
    """One event with a non-integer / negative / -0.0 / out-of-range coordinate anywhere in a window (any lane of any
    wave) must take the window off the pixel fast path; the result equals the general slicer's."""
    import os
    import torch
    ctx = env[0]
    from eventcalib_amd.pipeline import DetectPipeline
    buf = SS.make_stream(60000, rate=1.0e6, device="cpu", seed=9)
    rec = buf.numpy().reshape(-1, 25)
    t, _, _ = SS.unpack_records(buf)
    t0, t1 = SS.tiled_windows(float(t[0]), float(t[-1]))
    rng = np.random.default_rng(1)
    odd = [0.5, -3.0, -0.0, 2048.0, 1.0e300]
    for w in range(len(t0)):                      # one odd event per window, at a random position
        idx = np.nonzero((t.numpy() >= t0[w]) & (t.numpy() <= t1[w]))[0]
        if len(idx) == 0:
            continue
        k = int(rng.choice(idx))
        xy = rec[k, 8:24].copy().view(np.float64)
        xy[w % 2] = odd[w % len(odd)]
        rec[k, 8:24] = xy.view(np.uint8)
    ev = buf.cuda()
    outs = []
    for no_pixel in (False, True):
        if no_pixel:
            os.environ["ECAL_FORCE"] = "slice_general"
            __import__("eventcalib_amd.capi", fromlist=["sync_env"]).sync_env()   # (the switches are read once per context)
        try:
            p = DetectPipeline(ctx)
            p.set_windows(t0, t1)
            p.run(ev, slice_only=True)
            torch.cuda.synchronize()
            S = len(t0)
            outs.append((p.seg_off[:2 * S].cpu().numpy().copy(), p.seg_cnt[:2 * S].cpu().numpy().copy(),
                         p.event_point[:60000].cpu().numpy().copy(), p.xy[:60000].cpu().numpy().copy()))
        finally:
            os.environ.pop("ECAL_FORCE", None)
            __import__("eventcalib_amd.capi", fromlist=["sync_env"]).sync_env()   # (the switches are read once per context)
    a, b = outs
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    used = np.zeros(60000, bool)
    for o, c in zip(a[0], a[1]):
        used[o:o + c] = True
    assert np.array_equal(a[3][used].view(np.uint64), b[3][used].view(np.uint64))   # bitwise, -0.0 included


def test_hash_slicer_second_pass_equals_general_slicer(env):
    """Windows of 2048 ... 4095 events are taken by the second pass of the hash-table slicer (4096-slot tables, 12-bit event
    indices); its result must equal the general slicer's, and the oracle's on a sample of windows."""
    import os
    import torch
    ctx = env[0]
    from eventcalib_amd.pipeline import DetectPipeline
    n = 120000
    buf = SS.make_stream(n, rate=2.0e6, device="cpu", seed=21)
    t, _, _ = SS.unpack_records(buf)
    t0a, t1a = SS.tiled_windows(float(t[0]), float(t[-1]), 1.5e-3)      # ~3000 events
    t0b, t1b = SS.tiled_windows(float(t[0]), float(t[-1]), 2.0e-3)      # ~4000 events
    t0, t1 = np.concatenate([t0a, t0b]), np.concatenate([t1a, t1b])
    ev = buf.cuda()
    outs = []
    for no_pixel in (False, True):
        if no_pixel:
            os.environ["ECAL_FORCE"] = "slice_general"
            __import__("eventcalib_amd.capi", fromlist=["sync_env"]).sync_env()   # (the switches are read once per context)
        try:
            p = DetectPipeline(ctx)
            p.set_windows(t0, t1)
            p.run(ev, slots=2 * n + 8192, slice_only=True)
            torch.cuda.synchronize()
            S = len(t0)
            assert not p.overflowed()
            outs.append((p.win_lo[:S].cpu().numpy().copy(), p.win_hi[:S].cpu().numpy().copy(), p.seg_off[:2 * S].cpu().numpy().copy(),
                         p.seg_cnt[:2 * S].cpu().numpy().copy(), p.event_point.cpu().numpy().copy(), p.xy.cpu().numpy().copy(),
                         p.win_base[:S + 1].cpu().numpy().copy()))
        finally:
            os.environ.pop("ECAL_FORCE", None)
            __import__("eventcalib_amd.capi", fromlist=["sync_env"]).sync_env()   # (the switches are read once per context)
    a, b = outs
    assert int((a[1] - a[0]).max()) > 2048 and int((a[1] - a[0]).max()) < 4096
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    for s in range(len(t0)):
        lo, hi, base = int(a[0][s]), int(a[1][s]), int(a[6][s])
        assert np.array_equal(a[4][base:base + hi - lo], b[4][base:base + hi - lo]), s
        for pol in range(2):
            o, c = int(a[2][2 * s + pol]), int(a[3][2 * s + pol])
            assert np.array_equal(a[5][o:o + c], b[5][o:o + c]), (s, pol)


def test_hash_slicer_third_pass_equals_general_slicer(env):
    """Windows of 4096 ... 5119 events (what the keyframe search's windows of nine and ten steps hold) are taken by the THIRD pass of
    the hash-table slicer (8192-slot tables, 13-bit event indices, sets of up to 2357 keys per polarity); longer ones and sets
    beyond that go on to the general tiers.  Points, their order, the segments and the event -> point map must equal the general
    slicer's on every window (which tests above pin to the oracle)."""
    import os
    import torch
    ctx = env[0]
    from eventcalib_amd.pipeline import DetectPipeline
    n = 200000
    buf = SS.make_stream(n, rate=2.0e6, device="cpu", seed=33)
    t, _, _ = SS.unpack_records(buf)
    t0a, t1a = SS.tiled_windows(float(t[0]), float(t[-1]), 2.2e-3)      # ~4400 events
    t0b, t1b = SS.tiled_windows(float(t[0]), float(t[-1]), 2.5e-3)      # ~5000 events: some beyond 5119
    t0c, t1c = SS.tiled_windows(float(t[0]), float(t[-1]), 3.2e-3)      # ~6400 events: the general tiers'
    t0d, t1d = SS.tiled_windows(float(t[0]), float(t[-1]), 1.0e-3)      # ~2000 events: first and second pass beside them
    t0, t1 = np.concatenate([t0a, t0b, t0c, t0d]), np.concatenate([t1a, t1b, t1c, t1d])
    ev = buf.cuda()
    outs = []
    for no_pixel in (False, True):
        if no_pixel:
            os.environ["ECAL_FORCE"] = "slice_general"
            __import__("eventcalib_amd.capi", fromlist=["sync_env"]).sync_env()   # (the switches are read once per context)
        try:
            p = DetectPipeline(ctx)
            p.set_windows(t0, t1)
            p.run(ev, slots=int(4.2 * n) + 8192, slice_only=True)
            torch.cuda.synchronize()
            S = len(t0)
            assert not p.overflowed()
            fmt = p.seg_fmt[:2 * S].cpu().numpy().copy()   # (1: the window's points went out packed, i.e. through a hash pass)
            outs.append((p.win_lo[:S].cpu().numpy().copy(), p.win_hi[:S].cpu().numpy().copy(), p.seg_off[:2 * S].cpu().numpy().copy(),
                         p.seg_cnt[:2 * S].cpu().numpy().copy(), p.event_point.cpu().numpy().copy(), p.xy.cpu().numpy().copy(),
                         p.win_base[:S + 1].cpu().numpy().copy(), fmt))
        finally:
            os.environ.pop("ECAL_FORCE", None)
            __import__("eventcalib_amd.capi", fromlist=["sync_env"]).sync_env()   # (the switches are read once per context)
    a, b = outs
    sizes = a[1] - a[0]
    third = (sizes > 4095) & (sizes <= 5119)
    if ctx.point_order() == "reference":   # the third pass took its windows (a set of > 2357 keys may go on); first-occurrence order has none
        assert a[7][0::2][third].mean() > 0.9
    assert not a[7][0::2][sizes > 5119].any() and not b[7].any()
    assert int(((sizes > 4095) & (sizes <= 5119)).sum()) >= 40 and int((sizes > 5119).sum()) >= 10 and int((sizes <= 4095).sum()) >= 40
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    for s in range(len(t0)):
        lo, hi, base = int(a[0][s]), int(a[1][s]), int(a[6][s])
        assert np.array_equal(a[4][base:base + hi - lo], b[4][base:base + hi - lo]), s
        for pol in range(2):
            o, c = int(a[2][2 * s + pol]), int(a[3][2 * s + pol])
            assert np.array_equal(a[5][o:o + c], b[5][o:o + c]), (s, pol)


def test_third_pass_edges_equal_the_oracle(env):
    """The third hash pass at its limits, every window against the ORACLE (real std::unordered_set order): exactly 4096 / 5119 / 5120
    events; a set of exactly 2357 keys (the eighth epoch's bucket count: the last the pass holds) and of 2358 (a ninth epoch: the
    general tiers'); all events one pixel; x beyond the pass's 511 (general tiers); heavy +/- cancellation."""
    ctx, pipe, torch = env
    rng = np.random.default_rng(77)
    recs, bounds = [], []
    t_at = [0.0]

    def window(x, y, p):
        n = len(x)
        t = t_at[0] + 1e-6 * (1 + np.arange(n))
        recs.append(O.pack_events(t, np.asarray(x, np.float64), np.asarray(y, np.float64), np.asarray(p, np.uint8)))
        bounds.append((t[0], t[-1]))
        t_at[0] = t[-1] + 1e-3

    def distinct(n_keys, n_events, pol):   # n_keys distinct pixels (x <= 511), the rest duplicates of them
        pix = rng.choice(512 * 300, n_keys, replace=False)
        idx = np.concatenate([np.arange(n_keys), rng.integers(0, n_keys, n_events - n_keys)])
        idx[n_keys:] = idx[n_keys:][rng.permutation(n_events - n_keys)]
        return pix[idx] // 300, pix[idx] % 300, np.full(n_events, pol)

    for n in (4096, 5119, 5120):   # (a 60 x 50 patch: ~1700 keys per polarity, within the pass's 2357)
        window(rng.integers(100, 160, n), rng.integers(100, 150, n), rng.integers(0, 2, n))
    for keys in (2357, 2358):
        x, y, p = distinct(keys, 4500, 1)
        window(x, y, p)
        xa, ya, pa = distinct(keys, 2400, 0)
        xb, yb, pb = distinct(1500, 2300, 1)
        window(np.concatenate([xa, xb]), np.concatenate([ya, yb]), np.concatenate([pa, pb]))
    window(np.full(4700, 17), np.full(4700, 3), rng.integers(0, 2, 4700))
    x = rng.integers(0, 346, 4600)
    x[1234] = 600
    window(x, rng.integers(0, 260, 4600), rng.integers(0, 2, 4600))
    window(rng.integers(0, 40, 5000), rng.integers(0, 30, 5000), rng.integers(0, 2, 5000))
    rec = np.concatenate(recs)
    t0 = [b[0] for b in bounds]
    t1 = [b[1] + 5e-7 for b in bounds]
    pipe.set_windows(t0, t1)
    pipe.run(torch.from_numpy(rec).cuda(), slots=rec.size // 25 + 64)
    torch.cuda.synchronize()
    _compare(pipe, torch, rec, t0, t1, check_labels=False)
    if ctx.point_order() == "reference":
        fmt = pipe.seg_fmt[:2 * len(t0)].cpu().numpy()[0::2]
        # packed = taken by a hash pass: 4096, 5119 yes, 5120 no; 2357 keys yes (both windows), 2358 no; one pixel yes; x = 600 no; the last yes
        assert [int(v != 0) for v in fmt] == [1, 1, 0, 1, 1, 0, 0, 1, 0, 1]


def test_golden_eventframe_order_fixtures(env):
    """`.bin` records -> the reference's point order -> DBSCAN labels, against the committed fixtures
    (tests/golden/eventframe_order_*.npz: real std::unordered_set + the reference's kd-tree, made in the build container)."""
    import glob
    import os
    ctx, pipe, torch = env
    if ctx.point_order() != "reference":
        pytest.skip("the fixtures hold the reference's order")
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "eventframe_order_*.npz")))
    assert len(files) >= 4
    for f in files:
        g = np.load(f)
        rec = g["records"]
        t = rec.reshape(-1, 25)[:, :8].copy().view(np.float64).reshape(-1)
        b = g["bounds"]
        t0 = [float(t[lo]) for lo, hi in b]
        t1 = [float(t[hi - 1]) for lo, hi in b]
        S = len(b)
        pipe.set_windows(t0, t1)
        pipe.run(torch.from_numpy(rec).cuda(), slots=int((b[:, 1] - b[:, 0]).sum()) + 64, eps=float(g["eps"]), minpts=int(g["minpts"]),
                 detect=False)
        torch.cuda.synchronize()
        assert np.array_equal(pipe.win_lo[:S].cpu().numpy(), b[:, 0]) and np.array_equal(pipe.win_hi[:S].cpu().numpy(), b[:, 1]), f
        assert np.array_equal(pipe.seg_cnt[:2 * S].cpu().numpy(), g["seg_cnt"]), f
        assert np.array_equal(pipe.n_clusters[:2 * S].cpu().numpy(), g["n_clusters"]), f
        off = pipe.seg_off[:2 * S].cpu().numpy().astype(np.int64)
        base = pipe.win_base[:S + 1].cpu().numpy().astype(np.int64)
        xy, lab, ep = pipe.xy.cpu().numpy(), pipe.labels.cpu().numpy(), pipe.event_point.cpu().numpy()
        po, eo = 0, 0
        for s in range(S):
            for k in range(2):
                o, c = off[2 * s + k], int(g["seg_cnt"][2 * s + k])
                assert np.array_equal(xy[o:o + c].view(np.uint64), g["xy"][po:po + c].view(np.uint64)), (f, s, k)
                assert np.array_equal(lab[o:o + c], g["labels"][po:po + c]), (f, s, k)
                po += c
            n = int(b[s, 1] - b[s, 0])
            assert np.array_equal(ep[base[s]:base[s] + n], g["event_point"][eo:eo + n]), (f, s)
            eo += n


def test_unsorted_stream_is_sorted_like_the_multimap(env):
    """ecal_sort_events_dev / ecal_stream_create on records in arbitrary order: ascending time stamp, equal time stamps in
    input order (std::multimap::emplace, eventCameraCalib.cpp:154-163) == numpy's stable argsort; negative, zero (+-0.0) and
    repeated time stamps included."""
    ctx, pipe, torch = env
    rng = np.random.default_rng(8)
    n = 200_000
    t = rng.choice(np.concatenate([rng.uniform(-1, 3, 5000), [0.0, -0.0, 1.5, 1.5, 2.0]]), n)   # many equal keys
    x = np.arange(n, dtype=np.float64)             # the payload identifies the record
    rec = O.pack_events(t, x, x % 7, (np.arange(n) & 1).astype(np.uint8))
    d = torch.from_numpy(rec).cuda()
    out = torch.empty_like(d)
    ctx.sort_events_dev(d.data_ptr(), n, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want = rec.reshape(n, 25)[np.argsort(t, kind="stable")]
    assert np.array_equal(out.cpu().numpy().reshape(n, 25), want)
    # through the host-pointer entry point: the stream object holds the sorted records
    import ctypes
    L = ctx._L
    L.ecal_stream_create.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_void_p)]
    L.ecal_stream_create.restype = ctypes.c_int
    L.ecal_stream_data.argtypes = [ctypes.c_void_p]
    L.ecal_stream_data.restype = ctypes.c_void_p
    L.ecal_stream_destroy.argtypes = [ctypes.c_void_p]
    L.ecal_copy_dev.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
    h = ctypes.c_void_p()
    assert L.ecal_stream_create(ctx._h, rec.ctypes.data, n, ctypes.byref(h)) == 0
    back = torch.empty(n * 25, dtype=torch.uint8, device="cuda")
    assert L.ecal_copy_dev(ctx._h, back.data_ptr(), L.ecal_stream_data(h), n * 25, None, 1) == 0
    assert np.array_equal(back.cpu().numpy().reshape(n, 25), want)
    L.ecal_stream_destroy(h)


def test_event_point_map_is_optional():
    """d_event_point = NULL (include/ecal.h): the event -> point map is not written, everything else comes out the same — on
    the shipped configuration (reference order and first-occurrence order) and on windows of every tier."""
    import torch
    import eventcalib_amd
    import test_gpu_fused as TF
    from eventcalib_amd.pipeline import DetectPipeline
    ctx = eventcalib_amd.Context(0)
    try:
        n = 800_000
        ev = SS.make_stream(n, rate=2.0e6, device="cuda", seed=19, noise_frac=0.3)
        rng = np.random.default_rng(5)
        lens = rng.choice([0.0, 2e-5, 7e-4, 1.5e-3, 3e-3, 8e-3], size=200)
        starts = 5.0 + rng.uniform(0, n / 2.0e6 - 1e-2, size=200)
        for order in ("reference", "first"):
            ctx.set_point_order(order)
            snaps = []
            for want in (True, False):
                pipe = DetectPipeline(ctx, want_event_point=want)
                pipe.set_windows(starts, starts + lens)
                pipe._ensure(len(starts), 4_000_000)
                pipe.event_point.fill_(-77)
                pipe.run(ev, slots=4_000_000)
                torch.cuda.synchronize()
                assert not pipe.overflowed()
                snaps.append((TF._snapshot(pipe, len(starts), torch), pipe.event_point.clone()))
            (a, epa), (b, epb) = snaps
            assert bool((epb == -77).all()) and not bool((epa == -77).all())
            for k in a:
                if k == "event_point":
                    continue
                x, y = a[k], b[k]
                assert torch.equal(x.view(torch.int64) if x.dtype.is_floating_point else x, y.view(torch.int64) if y.dtype.is_floating_point else y), (order, k)
    finally:
        ctx.set_point_order("reference")
        ctx.close()
