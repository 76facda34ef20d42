"""Debug: does a large torch pinned copy before the keyframe search slow the search's passes?  (bench.py's h2d leg)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import eventcalib_amd, synth_stream as SS
from eventcalib_amd.adaptive import detect_keyframes_device
from eventcalib_amd.pipeline import DetectPipeline
n = 50_000_000
ctx = eventcalib_amd.Context(0)
ev = SS.make_stream(n, device="cuda")
pipe = DetectPipeline(ctx)
t0, t1 = SS.tiled_windows(5.0, 5.0 + (n - 1) / 1e6)
pipe.set_windows(t0, t1)
for _ in range(3):
    pipe.run(ev)
torch.cuda.synchronize()
def timed(tag):
    detect_keyframes_device(ctx, ev, 5e-4, 4000, 1270, 5.0, 6.0)
    torch.cuda.synchronize(); t = time.perf_counter()
    kf = detect_keyframes_device(ctx, ev, 5e-4, 4000, 1270, 5.0, 5.0 + (n - 1) / 1e6)
    torch.cuda.synchronize(); el = time.perf_counter() - t
    print(tag, "%.4f s" % el, len(kf["time"]), flush=True)
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
if mode == "pinned":
    host = torch.empty(ev.numel(), dtype=torch.uint8, pin_memory=True)
    host.copy_(ev)
    torch.cuda.synchronize()
    del host
elif mode == "dev":
    dst = torch.empty_like(ev); dst.copy_(ev); torch.cuda.synchronize(); del dst
elif mode == "small":
    pipe.set_windows(t0[:8000], t1[:8000]); pipe.run(ev); torch.cuda.synchronize()
timed(mode + " 1st"); timed(mode + " 2nd")
