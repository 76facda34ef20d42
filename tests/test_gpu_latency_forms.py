"""GPU: the stages' LATENCY forms (what the keyframe search runs its passes with: one launch takes a window / segment through the size
tier it needs, and on to the next tier inside the workgroup when that one cannot hold it) against the staged passes, outside the search
(ECAL_FORCE=latency_two_pass / latency_forms): slicing, DBSCAN and the exact extraction must give the same arrays on windows of every
tier, including the third hash pass's limits.  (The staged passes are the ones the other suites pin to the oracle.)"""
import os

import numpy as np
import pytest

import oracle_lib as O
import synth_stream as SS

pytestmark = pytest.mark.gpu


def _stream_and_windows():
    rng = np.random.default_rng(5)
    n = 160000
    buf = SS.make_stream(n, rate=2.0e6, device="cpu", seed=41).numpy()
    t, _, _ = SS.unpack_records(__import__("torch").from_numpy(buf))
    t0s, t1s = [], []
    for w in (0.8e-3, 1.3e-3, 1.9e-3, 2.3e-3, 2.5e-3, 3.3e-3):   # ~1600 ... 6600 events: every tier of every stage
        a, b = SS.tiled_windows(float(t[0]), float(t[-1]), w)
        t0s.append(a[::3])
        t1s.append(b[::3])
    # synthetic windows behind the stream: sets at the hash passes' key limits, one pixel, a coordinate beyond the third pass's range
    recs, t_at = [buf], float(t[-1]) + 1e-2
    extra0, extra1 = [], []

    def window(x, y, p):
        nonlocal t_at
        m = len(x)
        tt = t_at + 1e-6 * (1 + np.arange(m))
        recs.append(O.pack_events(tt, np.asarray(x, np.float64), np.asarray(y, np.float64), np.asarray(p, np.uint8)))
        extra0.append(tt[0])
        extra1.append(tt[-1] + 5e-7)
        t_at = tt[-1] + 1e-3

    def distinct(keys, events, pol):
        pix = rng.choice(512 * 300, keys, replace=False)
        idx = np.concatenate([np.arange(keys), rng.integers(0, keys, events - keys)])
        return pix[idx] // 300, pix[idx] % 300, np.full(events, pol)

    for keys, events in ((1109, 1800), (1110, 1800), (1400, 2000), (2048, 3900), (2049, 3900), (2357, 4500), (2358, 4500)):
        x, y, p = distinct(keys, events, 1)
        window(x, y, p)
    window(np.full(4700, 17), np.full(4700, 3), rng.integers(0, 2, 4700))
    x = rng.integers(0, 346, 4600)
    x[77] = 600
    window(x, rng.integers(0, 260, 4600), rng.integers(0, 2, 4600))
    x = rng.integers(0, 346, 1500).astype(np.float64)
    x[5] = 10.5
    window(x, rng.integers(0, 260, 1500), rng.integers(0, 2, 1500))
    rec = np.concatenate(recs)
    return rec, np.concatenate(t0s + [np.array(extra0)]), np.concatenate(t1s + [np.array(extra1)])


def _run(ctx, rec, t0, t1, force):
    import torch
    import eventcalib_amd.capi as capi
    from eventcalib_amd.pipeline import DetectPipeline
    if force:
        os.environ["ECAL_FORCE"] = force
    capi.sync_env()
    try:
        p = DetectPipeline(ctx)
        p.set_windows(t0, t1)
        n = rec.size // 25
        p.run(torch.from_numpy(rec).cuda(), slots=int(2.3 * n) + 8192)
        torch.cuda.synchronize()
        assert not p.overflowed()
        S = len(t0)
        out = {k: getattr(p, k)[:m].cpu().numpy().copy() for k, m in (("win_lo", S), ("win_hi", S), ("win_base", S + 1), ("seg_off", 2 * S),
                                                                      ("seg_cnt", 2 * S), ("n_clusters", 2 * S), ("win_info", S))}
        for k in ("xy", "event_point", "labels", "kept_labels", "rep", "cand_pair", "cand_xyr"):
            out[k] = getattr(p, k).cpu().numpy().copy()
        return out
    finally:
        os.environ.pop("ECAL_FORCE", None)
        capi.sync_env()


@pytest.mark.parametrize("force", ["latency_two_pass", "latency_forms"])
def test_latency_forms_equal_the_staged_passes(force):
    import eventcalib_amd
    ctx = eventcalib_amd.Context(0)
    try:
        rec, t0, t1 = _stream_and_windows()
        a = _run(ctx, rec, t0, t1, None)
        b = _run(ctx, rec, t0, t1, force)
        S = len(t0)
        sizes = a["win_hi"] - a["win_lo"]
        assert (sizes <= 2047).sum() >= 20 and ((sizes > 2047) & (sizes <= 4095)).sum() >= 20 and ((sizes > 4095) & (sizes <= 5119)).sum() >= 10 \
            and (sizes > 5119).sum() >= 5
        for k in ("win_lo", "win_hi", "win_base", "seg_off", "seg_cnt", "n_clusters"):
            assert np.array_equal(a[k], b[k]), k
        assert np.array_equal(a["win_info"], b["win_info"])
        for s in range(S):
            base, n = int(a["win_base"][s]), int(sizes[s])
            assert np.array_equal(a["event_point"][base:base + n], b["event_point"][base:base + n]), s
            for pol in range(2):
                o, c = int(a["seg_off"][2 * s + pol]), int(a["seg_cnt"][2 * s + pol])
                assert np.array_equal(a["xy"][o:o + c].view(np.uint64), b["xy"][o:o + c].view(np.uint64)), (s, pol)
                assert np.array_equal(a["labels"][o:o + c], b["labels"][o:o + c]), (s, pol)
            info = a["win_info"].reshape(S, -1)[s]
            nc = int(info[0])
            o = int(a["seg_off"][2 * s])
            assert np.array_equal(a["cand_xyr"][o:o + nc].view(np.uint64), b["cand_xyr"][o:o + nc].view(np.uint64)), s
            assert np.array_equal(a["cand_pair"][o:o + nc], b["cand_pair"][o:o + nc]), s
    finally:
        ctx.close()
