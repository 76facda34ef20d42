import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import eventcalib_amd, eventcalib_amd.capi as capi
from eventcalib_amd.adaptive import detect_keyframes_device
import synth_stream as SS
from test_gpu_adaptive import _ChainHandover
ctx = eventcalib_amd.Context(0)
ev = SS.make_stream(3_000_000, rate=2.0e6, device="cuda", seed=77)
torch.cuda.synchronize()
pieces, shards = 40, 40
t_first, t_last = 5.0, 5.4
want = detect_keyframes_device(ctx, ev, 5e-4, 4000, pieces, t_first, t_last, gate_mode=capi.GATE_SHARED_MAP)
step = (t_last - t_first) / pieces
for delay in (0, 2, None):
    frame = (0, 0.0, [0.0] * 64)
    for g in reversed(range(shards)):
        lo, hi = pieces * g // shards, pieces * (g + 1) // shards
        ho = _ChainHandover(frame, delay)
        p = detect_keyframes_device(ctx, ev, 5e-4, 4000, pieces, t_first, t_last, gate_mode=capi.GATE_SHARED_MAP, piece_first=lo, piece_count=hi - lo, handover=ho)
        t_lo, t_hi = t_last - step * hi, t_last - step * lo
        w = want["time"][(want["time"] >= t_lo) & (want["time"] < t_hi)]
        if not np.array_equal(p["time"], w):
            print("delay", delay, "shard", g, "pieces", lo, hi, "got", p["time"], "want", w, "polls", ho.polls, "frame in", frame[:2], "out", ho.frame_out[:2])
        frame = ho.frame_out
print("done")
