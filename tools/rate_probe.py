"""Debug: stage times of the tiled detection pass at other event rates / window lengths (how far off the pixel fast
paths a denser stream falls).  python tools/rate_probe.py [events] [rate] [window_s]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import eventcalib_amd
from eventcalib_amd.pipeline import DetectPipeline
import synth_stream as SS
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
rate = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0e6
wlen = float(sys.argv[3]) if len(sys.argv) > 3 else 1.5e-3
ctx = eventcalib_amd.Context(0); pipe = DetectPipeline(ctx)
ev = SS.make_stream(n, rate=rate, device="cuda")
t0, t1 = SS.tiled_windows(5.0, 5.0 + (n - 1) / rate, wlen)
pipe.set_windows(t0, t1)
for _ in range(2):
    pipe.run(ev)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    pipe.run(ev)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
S = len(t0)
print("rate %.1f Mev/s, window %.2f ms: %d events, %d windows, %.3f ms per pass = %.0f Mev/s; max segment %d points, windows reaching pairing %d"
      % (rate / 1e6, wlen * 1e3, n, S, ms, n / ms / 1e3, int(pipe.seg_cnt[:2 * S].max()), int((pipe.win_info[:S, 3] == 0).sum())))
