import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import eventcalib_amd, synth_stream as SS
from eventcalib_amd.pipeline import DetectPipeline
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ctx = eventcalib_amd.Context(0)
L = eventcalib_amd.load_library()
L.ecal_debug_tail_seen.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
ev = SS.make_stream(n, device="cuda")
pipe = DetectPipeline(ctx, want_event_point=False)
t0, t1 = SS.tiled_windows(5.0, 5.0 + (n - 1) / 1e6)
pipe.set_windows(t0, t1); pipe.set_detect_params(5, 36, 15.511363636363637)
for i in range(4):
    pipe.run(ev)
    torch.cuda.synchronize()
    b = (ctypes.c_uint32 * 16)()
    L.ecal_debug_tail_seen(ctx._h, b)
    print(i, [hex(v) if v > 1e6 else v for v in b][:10])
