"""The FIRST keyframe search of a process against the second (what the cold C++ driver's search stage pays over the warm one):
wall time of each + ECAL_TRACE=adaptive's scratch line.  python tools/cold_search_probe.py [n_events] [pieces]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if os.environ.get("COLD_PROBE_TRACE"): os.environ["ECAL_TRACE"] = "adaptive"
import torch
import eventcalib_amd, synth_stream as SS
import eventcalib_amd.capi as capi
from eventcalib_amd.adaptive import detect_keyframes_device
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
pieces = int(sys.argv[2]) if len(sys.argv) > 2 else 1270
ctx = eventcalib_amd.Context(0)
ev = SS.make_stream(n, device="cuda")
torch.cuda.synchronize()
if os.environ.get("COLD_PROBE_SPIN_MS"):   # a dummy load before the first search: is the first searches' extra time the clocks' ramp?
    a = torch.randn(4096, 4096, device="cuda")
    t = time.perf_counter()
    while (time.perf_counter() - t) * 1e3 < float(os.environ["COLD_PROBE_SPIN_MS"]):
        (a @ a).sum().item()
    print("spun %.0f ms" % ((time.perf_counter() - t) * 1e3), file=sys.stderr)
if os.environ.get("COLD_PROBE_LAUNCHES"):   # many cheap launches before the first search: is it the runtime's launch path that has to warm up?
    z = torch.zeros(64, device="cuda")
    t = time.perf_counter()
    for _ in range(int(os.environ["COLD_PROBE_LAUNCHES"])):
        z.add_(1.0)
    torch.cuda.synchronize()
    print("%s launches in %.1f ms" % (os.environ["COLD_PROBE_LAUNCHES"], (time.perf_counter() - t) * 1e3), file=sys.stderr)
for i in range(int(os.environ.get("COLD_PROBE_N", "3"))):
    t = time.perf_counter()
    kf = detect_keyframes_device(ctx, ev, 5e-4, 4000, pieces, 5.0, 5.0 + (n - 1) / 1e6, gate_mode=capi.GATE_SHARED_MAP)
    torch.cuda.synchronize()
    print("search %d: %.4f s, %d keyframes" % (i + 1, time.perf_counter() - t, len(kf["time"])), file=sys.stderr)
