"""Detection pass time with an alternative build of the library: python tools/ab_lib_pass.py path/to/libecal_x.so [events]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import eventcalib_amd.capi as capi
lib = sys.argv[1]
if lib != "-":
    capi.lib_path = lambda: os.path.abspath(lib)
import numpy as np, torch
import eventcalib_amd
from eventcalib_amd.pipeline import DetectPipeline
import synth_stream as SS
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000_000
ctx = eventcalib_amd.Context(0)
ev = SS.make_stream(n, device="cuda")
t0, t1 = SS.tiled_windows(5.0, 5.0 + (n - 1) / 1e6)
pipe = DetectPipeline(ctx)
pipe.set_windows(t0, t1)
pipe.set_detect_params(5, 36, 15.511363636363637)
for _ in range(3): pipe.run(ev)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): pipe.run(ev)
b.record(); torch.cuda.synchronize()
S = len(t0)
print("%s: pass %.3f ms; candidates %d fallback windows %d" % (lib, a.elapsed_time(b) / 10, int(pipe.win_info[:S, 0].sum()), int(((pipe.win_info[:S, 3] & 0x100) != 0).sum())))
