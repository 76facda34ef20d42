import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch, eventcalib_amd, synth_stream as SS
from eventcalib_amd.pipeline import DetectPipeline
from eventcalib_amd.adaptive import detect_keyframes_device
import eventcalib_amd.capi as capi
n = 50_000_000
SS.TRAJECTORY = "orbit"
ev = SS.make_stream(n, rate=1e6, t_start=5.0, device="cuda", seed=21)
ctx = eventcalib_amd.Context(0)
kf = detect_keyframes_device(ctx, ev, 5e-4, 4000, 1270, 5.0, 5.0 + (n - 1) / 1e6, gate_mode=capi.GATE_SHARED_MAP)
print(len(kf["time"]), "keyframes")
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pipe = DetectPipeline(ctx, ev.device)
    t1 = time.perf_counter()
    pipe.set_windows(kf["duration"][:, 0], kf["duration"][:, 1])
    torch.cuda.synchronize(); t2 = time.perf_counter()
    pipe.run(ev, 4.0, 2)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    _ = pipe.xy
    torch.cuda.synchronize(); t4 = time.perf_counter()
    print("new pipe %.2f ms, set_windows %.2f ms, run %.2f ms, xy %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
    del pipe
