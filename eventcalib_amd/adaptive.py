"""The reference driver's adaptive windowing + keyframe gate over a device-resident stream.

Python mirror of eventcalib_amd/csrc/host/multi_process.hpp (detect_keyframes), i.e. of
  MultiProcess::process     event_camera_calib/test/eventCameraCalib.cpp:34-97
  piece construction        event_camera_calib/test/eventCameraCalib.cpp:168-179
  EventCalibIni::track      event_camera_calib/src/EventCalibIni.cpp:23-97
Every piece advances in lockstep; one step = one batched pass of the detection pipeline (bounds -> slice ->
DBSCAN -> candidates -> grid order) over the current window of every active piece, then a small D2H copy of
the per-window verdicts drives the reference's success / slide / grow rule on the host.
"""
import numpy as np
import torch

from . import capi
from .pipeline import DetectPipeline


def _row_directions(feat, rows, cols):
    """feat [n, rows*cols, 2] -> unit directions [n, rows, 2] of the total-least-squares line through each
    pattern row (right singular vector of [x y 1] with the smallest singular value, EventCalibIni.cpp:46-57),
    oriented from the row's first to its last circle."""
    n = feat.shape[0]
    p = feat.reshape(n, rows, cols, 2)
    A = np.concatenate([p, np.ones((n, rows, cols, 1))], axis=3)
    _, _, vt = np.linalg.svd(A, full_matrices=False)
    line = vt[:, :, -1, :]                                   # (A, B, C)
    d = np.stack([line[:, :, 1], -line[:, :, 0]], axis=2)
    span = p[:, :, -1, :] - p[:, :, 0, :]
    sign = np.where((d * span).sum(axis=2) < 0, -1.0, 1.0)
    return d * sign[:, :, None]


def _nth_element_median(theta):
    """theta [n, rows]: what libstdc++'s std::nth_element leaves at position rows // 2 of every row — with NaNs among the
    angles (every comparison false) that is not the order statistic but whatever the library's loops leave there
    (EventCalibIni.cpp:78); the library's restatement is exported by libecal for exactly this (ecal_ref_nth_element_f64)."""
    import ctypes
    L = capi.load_library()
    L.ecal_ref_nth_element_f64.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32]
    L.ecal_ref_nth_element_f64.restype = None
    out = np.empty(theta.shape[0])
    for i in range(theta.shape[0]):
        row = np.ascontiguousarray(theta[i], np.float64).copy()
        L.ecal_ref_nth_element_f64(row.ctypes.data, row.shape[0], row.shape[0] // 2)
        out[i] = row[row.shape[0] // 2]
    return out


def orientation_gate(ref_feat, ref_t, cur_feat, cur_t, rows, cols, motion_time_step):
    """EventCalibIni::track's test for n frame pairs: median angle between corresponding pattern rows divided by the
    time distance below (5e-4 pi) / MotionTimeStep rad/s."""
    a, b = _row_directions(ref_feat, rows, cols), _row_directions(cur_feat, rows, cols)
    # Eigen's norm(): sqrt of the plain sum of squares (np.linalg.norm may scale; one ulp decides 0 against NaN below)
    c = (a * b).sum(axis=2) / (np.sqrt(a[..., 0] * a[..., 0] + a[..., 1] * a[..., 1]) * np.sqrt(b[..., 0] * b[..., 0] + b[..., 1] * b[..., 1]))
    with np.errstate(invalid="ignore"):
        theta = np.arccos(c)         # (not clamped, as the reference: a cosine rounded above 1 gives NaN)
        med = _nth_element_median(theta)
        return med / np.abs(cur_t - ref_t) < (5e-4 * np.pi) / motion_time_step


def detect_keyframes(pipe: DetectPipeline, events, motion_time_step, frame_event_num_threshold, piece_num,
                     start_time, end_time, eps=4.0, minpts=2, rows=9, cols=4, max_steps=1_000_000, n_threads=1):
    """Returns dict(time [K], duration [K,2], events_num [K], features [K, rows*cols, 3]) sorted by time, plus
    `steps` and `windows` (how many batched passes / windows were evaluated).

    n_threads > 1: the pieces are dealt round-robin to that many host threads, each with its OWN ecal_ctx (own stream
    and scratch — the ABI's one-context-per-thread rule, as the reference gives every worker its own DBSCAN instance):
    a pass is a ~0.4 ms chain of small launches that leaves the GPU mostly idle, so several chains run side by side.
    The result does not depend on n_threads (pieces are independent)."""
    torch.cuda.synchronize(events.device)   # the passes run on the context's own stream: `events` must be complete
    if n_threads > 1:
        import threading
        from .capi import Context
        ctxs = [Context(pipe.ctx.device) for _ in range(n_threads)]
        outs = [None] * n_threads

        def work(t):
            outs[t] = _detect_pieces(ctxs[t], events, motion_time_step, frame_event_num_threshold, piece_num,
                                     np.arange(t, piece_num, n_threads), start_time, end_time, eps, minpts, rows, cols, max_steps)
        th = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        for c in ctxs:
            c.close()
        keys = [r for o in outs for r in o[0]]
        steps, windows = max(o[1] for o in outs), sum(o[2] for o in outs)
    else:
        keys, steps, windows = _detect_pieces(pipe.ctx, events, motion_time_step, frame_event_num_threshold, piece_num,
                                              np.arange(piece_num), start_time, end_time, eps, minpts, rows, cols, max_steps)
    n = rows * cols
    keys.sort(key=lambda r: r[0])
    K = len(keys)
    return dict(time=np.array([r[0] for r in keys]), duration=np.array([[r[1], r[2]] for r in keys]).reshape(K, 2),
                events_num=np.array([r[3] for r in keys], np.int64),
                features=np.stack([r[4] for r in keys]) if K else np.zeros((0, n, 3)), steps=steps, windows=windows)


def detect_keyframes_device(ctx, events, motion_time_step, frame_event_num_threshold, piece_num, start_time, end_time, eps=4.0,
                            minpts=2, rows=9, cols=4, max_passes=0, gate_mode=0, n_threads=1, contexts=None, piece_first=0,
                            piece_count=0, handover=None):
    """Same result as detect_keyframes, with the policy on the device (ecal_detect_keyframes): no per-pass host round trip.
    gate_mode = capi.GATE_SHARED_MAP: the reference's single-worker run (one keyframe map for all pieces) instead of the
    own-piece gate.  Slots and the keyframe capacity are estimated and doubled when the library reports them too small.

    n_threads > 1 (own-piece gate only): the pieces are cut into that many contiguous groups, each searched by its own call
    on its own context and host thread (ecal_adaptive_params.piece_first / piece_count) — a lock-step pass is a chain of
    latency-bound launches; the keyframes are the same ones (measured on one MI355X: no gain, the passes of 1270 pieces
    keep the GPU busy — the cut is there for several GPUs).  `contexts`: the contexts to use (kept by the caller between
    calls: their scratch buffers stay allocated), else created and closed here.

    piece_count != 0: only the pieces piece_first .. piece_first + piece_count - 1 (one rank's share of a search cut over
    GPUs; `events` then only has to hold those pieces' time range).  Under capi.GATE_SHARED_MAP such a share needs the frame of
    the pieces before it: handover (DistHandover below; ecal_detect_keyframes_sharded)."""
    torch.cuda.synchronize(events.device)   # the passes run on the contexts' own streams: `events` must be complete
    n_ev = events.numel() // 25
    n_threads = max(1, min(int(n_threads), int(piece_num)))
    if n_threads == 1 or gate_mode != 0 or piece_count:
        return _detect_group(ctx, events, n_ev, motion_time_step, frame_event_num_threshold, piece_num, start_time, end_time, eps, minpts,
                             rows, cols, max_passes, gate_mode, piece_first, piece_count, handover)
    import threading
    from .capi import Context
    own = contexts is None
    ctxs = [Context(ctx.device) for _ in range(n_threads)] if own else list(contexts)[:n_threads]
    cuts = [piece_num * g // n_threads for g in range(n_threads + 1)]
    outs, errs = [None] * n_threads, [None] * n_threads

    def work(g):
        try:
            outs[g] = _detect_group(ctxs[g], events, n_ev, motion_time_step, frame_event_num_threshold, piece_num, start_time, end_time,
                                    eps, minpts, rows, cols, max_passes, gate_mode, cuts[g], cuts[g + 1] - cuts[g])
        except BaseException as e:   # noqa: BLE001 — re-raised in the caller's thread
            errs[g] = e
    th = [threading.Thread(target=work, args=(g,)) for g in range(n_threads)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    if own:
        for c in ctxs:
            c.close()
    for e in errs:
        if e is not None:
            raise e
    t = np.concatenate([o["time"] for o in outs])
    d = np.concatenate([o["duration"] for o in outs])
    o_ = np.lexsort((d[:, 0], t))          # the library's order: time stamp, then window start
    return dict(time=t[o_], duration=d[o_], events_num=np.concatenate([o["events_num"] for o in outs])[o_],
                features=np.concatenate([o["features"] for o in outs])[o_], steps=max(o["steps"] for o in outs),
                windows=sum(o["windows"] for o in outs))


class DistHandover:
    """The frame hand-over of a shared-map search cut over torch.distributed ranks in time (ecal_detect_keyframes_sharded): `src` =
    the rank that holds the pieces BEFORE this rank's in time (None: this rank holds the run's first piece), `dst` = the rank that
    holds the pieces after (None: the last).  One message of 2 + 64 doubles per boundary; the received frame is kept, so a call
    repeated with more capacity (ECAL_ERR_RANGE) finds it again and sends its own only once."""

    def __init__(self, src, dst, tag=0):
        import torch.distributed as dist
        self.dist, self.src, self.dst, self.tag = dist, src, dst, tag
        self.dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        self.frame = self.work = self.buf = None
        self.sent = False

    def recv(self, wait):
        if self.frame is None:
            if self.src is None:
                return (0, 0.0, [0.0] * 64)
            if self.work is None:
                self.buf = torch.zeros(66, dtype=torch.float64, device=self.dev)
                self.work = self.dist.irecv(self.buf, src=self.src, tag=self.tag)
            if wait:
                self.work.wait()
            elif not self.work.is_completed():
                return None
            b = self.buf.cpu().tolist()
            self.frame = (int(b[0]), b[1], b[2:])
        return self.frame

    def send(self, has, time, dirs):
        if self.sent or self.dst is None:
            return
        self.sent = True
        t = torch.tensor([float(has), float(time)] + list(dirs) + [0.0] * (64 - len(dirs)), dtype=torch.float64, device=self.dev)
        self.dist.send(t, dst=self.dst, tag=self.tag)


def _detect_group(ctx, events, n_ev, motion_time_step, frame_event_num_threshold, piece_num, start_time, end_time, eps, minpts, rows,
                  cols, max_passes, gate_mode, piece_first, piece_count, handover=None):
    span = max(end_time - start_time, 1e-9)
    n_mine = piece_count if piece_count else piece_num
    # a pass covers a chain of windows per piece: the library's own estimate, doubled whenever it reports it too small
    cap_max = min(2 ** 32 - 64, 6 * n_ev + 4096)     # (the hint's own maximum: the windows of one slot index are disjoint)
    cap = capi.detect_keyframes_cap_hint_dev(ctx, events.data_ptr(), n_ev, motion_time_step, frame_event_num_threshold, piece_num,
                                             start_time, end_time, piece_first, piece_count)
    if cap == 0:
        raise capi.EcalError(-1, "ecal_detect_keyframes_cap_hint_dev: invalid parameters")
    max_keys = int(span * n_mine / piece_num / (8 * motion_time_step)) + n_mine + 64   # one keyframe per window + gap at the very most
    while True:
        try:
            t, d, e, f, passes, windows = capi.detect_keyframes_dev(ctx, events.data_ptr(), n_ev, motion_time_step,
                                                                    frame_event_num_threshold, piece_num, start_time, end_time, cap,
                                                                    max_keys, eps, minpts, 5, rows, cols, max_passes=max_passes,
                                                                    gate_mode=gate_mode, piece_first=piece_first,
                                                                    piece_count=piece_count, handover=handover)
            break
        except capi.EcalError as err:
            if err.status != -6:
                raise
            # double only what was short: the call hands back the keyframe count with the error, and a count beyond the
            # capacity is what "max_keyframes too small" means (no reading of the message text)
            if getattr(err, "n_keyframes", 0) > max_keys:
                if max_keys > 4 * n_ev:
                    raise
                max_keys = max(2 * max_keys, int(err.n_keyframes) + 64)
            else:
                if cap >= cap_max:
                    raise
                cap = min(cap_max, 2 * cap)
    return dict(time=t, duration=d, events_num=e, features=f, steps=passes, windows=windows)


def _detect_pieces(ctx, events, motion_time_step, frame_event_num_threshold, piece_num, which, start_time, end_time, eps, minpts,
                   rows, cols, max_steps):
    """The lock-step loop over the pieces `which` (indices into the piece_num pieces of [start_time, end_time])."""
    n_ev = events.numel() // 25
    ln, gap = 3 * motion_time_step, 5 * motion_time_step
    # windows grow to at most 10 steps: twice the mean event count of such a span, to start with
    per_window = int(2 * 10 * motion_time_step * n_ev / max(end_time - start_time, 1e-9)) + 1024
    step = (end_time - start_time) / piece_num
    k = np.asarray(which)
    piece_num = len(k)
    bound_hi = end_time - step * k
    first = end_time - step * (k + 1)
    second = first + ln
    active = second < bound_hi
    have_ref = np.zeros(piece_num, bool)
    ref_feat = np.zeros((piece_num, rows * cols, 2))
    ref_t = np.zeros(piece_num)
    keys = []
    n = rows * cols
    steps = windows = 0
    while active.any() and steps < max_steps:
        idx = np.nonzero(active)[0]
        S = len(idx)
        # one C call per pass (ecal_detect_pass): bounds upload, five stages + grid ordering, one packed download
        while True:   # slots for the pass: windows x (a bound on the events of one window); doubled if a window was denser
            try:
                packed = capi.detect_pass(ctx, events.data_ptr(), n_ev, first[idx], second[idx], min(n_ev, S * per_window), eps,
                                          minpts, 5, rows, cols)
                break
            except capi.EcalError as e:
                if e.status != -6 or S * per_window >= n_ev:
                    raise
                per_window *= 2
        status, found, cnt = packed[:, 0].astype(np.int64) & 0xFF, packed[:, 1] != 0, packed[:, 2].astype(np.int64)   # cnt = EventFrame::eventsNum()
        ok = (status == 0) & found
        accepted = np.zeros(S, bool)
        if ok.any():
            w = np.nonzero(ok)[0]
            feat = packed[w, 3:].reshape(len(w), n, 3)
            t_mid = (first[idx[w]] + second[idx[w]]) / 2
            pw = idx[w]
            acc = ~have_ref[pw]
            chk = np.nonzero(have_ref[pw])[0]
            if len(chk):
                acc[chk] = orientation_gate(ref_feat[pw[chk]], ref_t[pw[chk]], feat[chk, :, :2], t_mid[chk], rows, cols,
                                            motion_time_step)
            for j in np.nonzero(acc)[0]:
                keys.append((t_mid[j], first[pw[j]], second[pw[j]], int(cnt[w[j]]), feat[j]))
            a = pw[acc]
            have_ref[a] = True
            ref_feat[a] = feat[acc, :, :2]
            ref_t[a] = t_mid[acc]
            accepted[w[acc]] = True
        # eventCameraCalib.cpp:61-62 (success), :67-69,75-77 (slide), :70-71,78-79 (grow)
        f, s2 = first[idx], second[idx]
        slide = ~accepted & ((cnt > frame_event_num_threshold) | ((s2 - f) > 3 * ln))
        grow = ~accepted & ~slide
        nf = np.where(accepted, s2 + gap, np.where(slide, f + motion_time_step, f))
        ns = np.where(grow, s2 + motion_time_step, nf + ln)
        first[idx], second[idx] = nf, ns
        active[idx] = ns < bound_hi[idx]
        steps += 1
        windows += S
    return keys, steps, windows
