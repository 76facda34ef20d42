"""GPU: the adaptive-window driver with the policy on the device (ecal_detect_keyframes) against the host-driven
lock-step driver (eventcalib_amd.adaptive.detect_keyframes over ecal_detect_pass): same keyframes, same windows."""
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
    ev = SS.make_stream(3_000_000, rate=2.0e6, device="cuda", seed=77)
    torch.cuda.synchronize()   # the library works on its own stream
    yield ctx, DetectPipeline(ctx), ev, torch
    ctx.close()


@pytest.mark.parametrize("pieces", [1, 7, 64, 500])
def test_device_policy_equals_host_policy(env, pieces):
    from eventcalib_amd.adaptive import detect_keyframes, detect_keyframes_device
    ctx, pipe, ev, torch = env
    t_first, t_last = 5.0, 5.0 + (3_000_000 - 1) / 2.0e6
    host = detect_keyframes(pipe, ev, 5e-4, 4000, pieces, t_first, t_last)
    dev = detect_keyframes_device(ctx, ev, 5e-4, 4000, pieces, t_first, t_last)
    assert dev["steps"] == host["steps"] and dev["windows"] == host["windows"]
    assert len(host["time"]) >= 5 or pieces == 1
    # the gate compares an angle rate with a threshold; host (numpy SVD) and device (Jacobi) agree to ~1e-13 in the angle,
    # so the accepted set is the same unless a frame sits on the threshold (none does on this stream)
    assert np.array_equal(dev["time"], host["time"])
    assert np.array_equal(dev["duration"], host["duration"])
    assert np.array_equal(dev["events_num"], host["events_num"])
    assert np.array_equal(dev["features"], host["features"])      # the same candidate circles, bit for bit


@pytest.mark.parametrize("pieces,groups", [(7, 2), (64, 3), (500, 4)])
def test_piece_subsets_add_up_to_the_whole_search(env, pieces, groups):
    """ecal_adaptive_params.piece_first / piece_count: a call searches only some of the pieces, with the bounds they have in the
    whole run (how the search is cut over contexts / host threads and over GPUs).  The union of the groups' keyframes is the
    whole run's, bit for bit; the shared-map gate refuses a subset (a piece's gate frame comes from the pieces before it)."""
    from eventcalib_amd import capi
    from eventcalib_amd.adaptive import detect_keyframes_device
    ctx, pipe, ev, torch = env
    t_first, t_last = 5.0, 5.0 + (3_000_000 - 1) / 2.0e6
    whole = detect_keyframes_device(ctx, ev, 5e-4, 4000, pieces, t_first, t_last)
    parts = detect_keyframes_device(ctx, ev, 5e-4, 4000, pieces, t_first, t_last, n_threads=groups)
    assert len(whole["time"]) >= 5
    for k in ("time", "duration", "events_num", "features"):
        assert np.array_equal(parts[k], whole[k]), k
    assert parts["windows"] == whole["windows"] and parts["steps"] == whole["steps"]
    n_ev = ev.numel() // 25
    with pytest.raises(capi.EcalError) as e:
        capi.detect_keyframes_dev(ctx, ev.data_ptr(), n_ev, 5e-4, 4000, pieces, t_first, t_last, 1 << 22, 4096,
                                  gate_mode=capi.GATE_SHARED_MAP, piece_first=1, piece_count=pieces - 1)
    assert e.value.status == -1
    with pytest.raises(capi.EcalError) as e:
        capi.detect_keyframes_dev(ctx, ev.data_ptr(), n_ev, 5e-4, 4000, pieces, t_first, t_last, 1 << 22, 4096, piece_first=2,
                                  piece_count=pieces - 1)
    assert e.value.status == -1


def test_capacity_errors_and_limits(env):
    import eventcalib_amd.capi as capi
    ctx, pipe, ev, torch = env
    n = ev.numel() // 25
    with pytest.raises(capi.EcalError) as e:   # slots for one pass too small
        capi.detect_keyframes_dev(ctx, ev.data_ptr(), n, 5e-4, 4000, 64, 5.0, 6.0, 1000, 4096)
    assert e.value.status == -6
    with pytest.raises(capi.EcalError) as e:   # keyframe capacity too small
        capi.detect_keyframes_dev(ctx, ev.data_ptr(), n, 5e-4, 4000, 64, 5.0, 6.0, n, 2)
    assert e.value.status == -6
    t, d, e_, f, passes, windows = capi.detect_keyframes_dev(ctx, ev.data_ptr(), n, 5e-4, 4000, 64, 5.0, 6.0, n, 4096, max_passes=3)
    assert passes <= 3 and windows <= 3 * 64
    with pytest.raises(capi.EcalError):        # invalid: no pieces
        capi.detect_keyframes_dev(ctx, ev.data_ptr(), n, 5e-4, 4000, 0, 5.0, 6.0, n, 16)


@pytest.mark.parametrize("pieces", [1, 6, 40])
def test_device_policy_equals_the_policy_oracle(env, pieces):
    """ecal_detect_keyframes against oracle/policy_oracle.cpp — the sequential loop of MultiProcess::process
    (eventCameraCalib.cpp:34-97) with TrackingBase::process + EventCalibIni::track (EventCalibIni.cpp:23-97) restated on the
    CPU, own-piece gate.  extractFeatures() of a window is supplied to the oracle by a callback that runs the product's
    detection stages on that single window (those stages have their own oracles): what is checked here is the control flow —
    window arithmetic in the reference's floating-point order, success / slide / grow, the strict piece end — and the gate."""
    import oracle_lib as O
    import eventcalib_amd.capi as capi
    from eventcalib_amd.adaptive import detect_keyframes_device
    ctx, pipe, ev, torch = env
    n_ev = ev.numel() // 25
    t_first, t_last = 5.0, 5.0 + 0.12                  # 240 k events of the 2 Mev/s stream
    calls = []

    def detect(t0, t1):
        packed = capi.detect_pass(ctx, ev.data_ptr(), n_ev, np.array([t0]), np.array([t1]), 65536, 4.0, 2, 5, 9, 4)
        calls.append((t0, t1))
        found = packed[0, 0] == 0 and packed[0, 1] != 0
        return found, int(packed[0, 2]), packed[0, 3:].reshape(36, 3) if found else None
    ref = O.policy_run(detect, t_first, t_last, pieces, 5e-4, 4000, 9, 4, mode=0)
    dev = detect_keyframes_device(ctx, ev, 5e-4, 4000, pieces, t_first, t_last)
    assert len(ref["time"]) >= 8
    assert dev["windows"] == ref["windows"]
    assert np.array_equal(dev["time"], ref["time"])
    assert np.array_equal(dev["duration"], ref["duration"])
    assert np.array_equal(dev["events_num"], ref["events_num"])
    assert np.array_equal(dev["features"], ref["features"])


@pytest.mark.parametrize("slots,chain", [(1, 1), (2, 2), (3, 64), (6, 5)])
def test_look_ahead_does_not_change_the_keyframes(env, slots, chain, monkeypatch):
    """How far a pass looks ahead along a piece's likely chain of windows (window slots per piece, longest chain: debug
    switches of ecal_detect_keyframes; (1, 1) = one window per piece and pass, the reference's loop as it stands) decides
    the number of passes, never the result."""
    from eventcalib_amd.adaptive import detect_keyframes_device
    ctx, pipe, ev, torch = env
    t_first, t_last = 5.0, 5.0 + 0.4
    want = detect_keyframes_device(ctx, ev, 5e-4, 4000, 23, t_first, t_last)
    monkeypatch.setenv("ECAL_ADAPTIVE_SHAPE", "depth=%d,depth_max=%d" % (slots, chain))
    __import__("eventcalib_amd.capi", fromlist=["sync_env"]).sync_env()   # (the switches are read once per context)
    got = detect_keyframes_device(ctx, ev, 5e-4, 4000, 23, t_first, t_last)
    assert len(want["time"]) >= 20
    for k in ("time", "duration", "events_num", "features"):
        assert np.array_equal(got[k], want[k]), k
    assert got["steps"] == want["steps"] and got["windows"] == want["windows"]


def _oracle_detect(ctx, ev, n_ev, cache):
    import eventcalib_amd.capi as capi

    def detect(t0, t1):
        if (t0, t1) not in cache:
            packed = capi.detect_pass(ctx, ev.data_ptr(), n_ev, np.array([t0]), np.array([t1]), 65536, 4.0, 2, 5, 9, 4)
            found = (int(packed[0, 0]) & 0xFF) == 0 and packed[0, 1] != 0
            cache[(t0, t1)] = (found, int(packed[0, 2]), packed[0, 3:].reshape(36, 3).copy() if found else None)
        return cache[(t0, t1)]
    return detect


def _same_keyframes(dev, ref):
    assert dev["windows"] == ref["windows"]
    for k in ("time", "duration", "events_num", "features"):
        assert np.array_equal(dev[k], ref[k]), k


def _in_rounds(ctx, ev, pieces, t_first, t_last, switch, value):
    """The shared-map search WITHOUT the side chains behind accepted windows (ECAL_ADAPTIVE_SHAPE side=0; with them is the default since
    round 5), or with another shape of the tree of chains a piece's window slots form (ECAL_ADAPTIVE_SHAPE tree=…; 0: none) — must give the
    same keyframes."""
    import os
    import eventcalib_amd.capi as capi
    from eventcalib_amd.adaptive import detect_keyframes_device
    os.environ["ECAL_ADAPTIVE_SHAPE"] = "%s=%s" % (switch, value)
    capi.sync_env()
    try:
        return detect_keyframes_device(ctx, ev, 5e-4, 4000, pieces, t_first, t_last, gate_mode=capi.GATE_SHARED_MAP)
    finally:
        del os.environ["ECAL_ADAPTIVE_SHAPE"]
        capi.sync_env()


@pytest.mark.parametrize("pieces", [1, 6, 40])
def test_shared_map_gate_equals_the_single_worker_reference(env, pieces):
    """gate_mode = ECAL_GATE_SHARED_MAP against oracle/policy_oracle.cpp mode 1: ONE keyframe map for all pieces, pieces in the
    reference's pop_back order, only the very first frame ungated (TrackingBase.cpp:16-46, EventCalibIni.cpp:26-36) — what the
    reference computes with a single worker thread.  The product gets there by speculation + verification, pass by pass
    (ecal_adaptive.hip: adaptive_verify_live_kernel); keyframes, windows and counts must be the sequential run's."""
    import oracle_lib as O
    import eventcalib_amd.capi as capi
    from eventcalib_amd.adaptive import detect_keyframes_device
    ctx, pipe, ev, torch = env
    n_ev = ev.numel() // 25
    t_first, t_last = 5.0, 5.0 + 0.4
    cache = {}
    detect = _oracle_detect(ctx, ev, n_ev, cache)
    ref = O.policy_run(detect, t_first, t_last, pieces, 5e-4, 4000, 9, 4, mode=1)
    own = O.policy_run(detect, t_first, t_last, pieces, 5e-4, 4000, 9, 4, mode=0)
    dev = detect_keyframes_device(ctx, ev, 5e-4, 4000, pieces, t_first, t_last, gate_mode=capi.GATE_SHARED_MAP)
    assert len(ref["time"]) >= 20
    _same_keyframes(dev, ref)
    _same_keyframes(_in_rounds(ctx, ev, pieces, t_first, t_last, switch="side", value="0"), ref)   # without the side chains behind accepted windows
    _same_keyframes(_in_rounds(ctx, ev, pieces, t_first, t_last, switch="side", value="1"), ref)   # the measured layout, named
    _same_keyframes(_in_rounds(ctx, ev, pieces, t_first, t_last, switch="tree", value="0"), ref)   # chains and side chains only (no tree of chains behind acceptances)
    _same_keyframes(_in_rounds(ctx, ev, pieces, t_first, t_last, switch="tree", value=str(4 | (4 << 8) | (2 << 16) | (2 << 20) | (2 << 24) | (1 << 28))), ref)   # a small tree: chains that end early, three levels
    _same_keyframes(_in_rounds(ctx, ev, pieces, t_first, t_last, switch="tree", value=str(12 | (10 << 8) | (6 << 16) | (5 << 20) | (0 << 24) | (0 << 28))), ref)   # a wide one from position 0, two levels
    if pieces == 1:
        _same_keyframes(dev, own)              # one piece: the two modes are the same run
    _same_keyframes(detect_keyframes_device(ctx, ev, 5e-4, 4000, pieces, t_first, t_last), own)


def test_shared_map_gate_at_the_reference_s_piece_count():
    """1270 pieces (5 x (256 - 2) hardware threads, eventCameraCalib.cpp:172-173) over a 12.7 s stream: pieces of 10 ms, most of
    them starting within a few milliseconds of their predecessor's last keyframe — the gate across the piece boundary
    rejects many first successes, the verification runs several rounds.  == the sequential single-worker oracle."""
    import torch
    import eventcalib_amd
    import eventcalib_amd.capi as capi
    import oracle_lib as O
    from eventcalib_amd.adaptive import detect_keyframes_device
    ctx = eventcalib_amd.Context(0)
    try:
        n_ev = 12_700_000
        ev = SS.make_stream(n_ev, device="cuda", seed=5)
        torch.cuda.synchronize()
        t_first, t_last = 5.0, 5.0 + (n_ev - 1) / 1e6
        cache = {}
        detect = _oracle_detect(ctx, ev, n_ev, cache)
        ref = O.policy_run(detect, t_first, t_last, 1270, 5e-4, 4000, 9, 4, mode=1)
        own = O.policy_run(detect, t_first, t_last, 1270, 5e-4, 4000, 9, 4, mode=0)
        dev = detect_keyframes_device(ctx, ev, 5e-4, 4000, 1270, t_first, t_last, gate_mode=capi.GATE_SHARED_MAP)
        print("keyframes: shared map %d, own piece %d" % (len(ref["time"]), len(own["time"])))
        assert len(ref["time"]) >= 1000 and not np.array_equal(ref["time"], own["time"])   # the modes differ: the test bites
        _same_keyframes(dev, ref)
        _same_keyframes(_in_rounds(ctx, ev, 1270, t_first, t_last, switch="tree", value="0"), ref)
        _same_keyframes(detect_keyframes_device(ctx, ev, 5e-4, 4000, 1270, t_first, t_last), own)
    finally:
        ctx.close()


class _ChainHandover:
    """In-process stand-in for the ranks' frame hand-over: the shards run one after the other, earliest in time first; `delay` =
    polls (wait = False) answered "not there yet" before the frame is handed out (None: only a waiting recv gets it)."""

    def __init__(self, frame_in, delay):
        self.frame_in, self.delay, self.polls, self.frame_out = frame_in, delay, 0, None

    def recv(self, wait):
        if wait or (self.delay is not None and self.polls >= self.delay):
            return self.frame_in
        self.polls += 1
        return None

    def send(self, has, time, dirs):
        self.frame_out = (has, time, dirs)


@pytest.mark.parametrize("pieces,shards,delay", [(40, 4, 0), (40, 4, 3), (40, 8, None), (23, 2, 1), (40, 40, 2)])
def test_sharded_shared_map_search_equals_the_single_call(env, pieces, shards, delay):
    """ecal_detect_keyframes_sharded: the pieces of ONE shared-map search cut into `shards` contiguous groups (as `bench.py --gpus N`
    cuts them over the ranks), every group a call of its own that gets the frame behind the earlier groups through the hand-over
    callbacks — immediately, after a few passes (the group has been running as a speculation without a frame: the pass-by-pass
    verification re-runs what the frame changes), or only when it has nothing left to do and waits.  The union of the groups'
    keyframes == the single call over all pieces == the sequential oracle (test_shared_map_gate_equals_the_single_worker_reference)."""
    import eventcalib_amd.capi as capi
    from eventcalib_amd.adaptive import detect_keyframes_device
    ctx, pipe, ev, torch = env
    t_first, t_last = 5.0, 5.0 + 0.4
    want = detect_keyframes_device(ctx, ev, 5e-4, 4000, pieces, t_first, t_last, gate_mode=capi.GATE_SHARED_MAP)
    assert len(want["time"]) >= 20
    frame = (0, 0.0, [0.0] * 64)
    parts = []
    for g in reversed(range(shards)):               # group `shards - 1` holds the largest piece indices = the earliest pieces
        lo, hi = pieces * g // shards, pieces * (g + 1) // shards
        ho = _ChainHandover(frame, delay)
        parts.append(detect_keyframes_device(ctx, ev, 5e-4, 4000, pieces, t_first, t_last, gate_mode=capi.GATE_SHARED_MAP,
                                             piece_first=lo, piece_count=hi - lo, handover=ho))
        assert ho.frame_out is not None
        if hi < pieces and delay is not None and delay > 0:
            assert ho.polls == delay                 # (the frame was withheld for that many passes)
        frame = ho.frame_out
    got = {k: np.concatenate([p[k] for p in parts]) for k in ("time", "duration", "events_num", "features")}
    order = np.argsort(got["time"], kind="stable")
    for k in ("time", "duration", "events_num", "features"):
        assert np.array_equal(got[k][order], want[k]), k
    assert sum(p["windows"] for p in parts) == want["windows"]
    # the frame behind the whole run = the last keyframe's time stamp
    assert frame[0] == 1 and frame[1] == want["time"][-1]
    # a subset without the hand-over stays refused
    with pytest.raises(capi.EcalError):
        detect_keyframes_device(ctx, ev, 5e-4, 4000, pieces, t_first, t_last, gate_mode=capi.GATE_SHARED_MAP, piece_first=0, piece_count=1)
