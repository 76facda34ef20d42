"""A/B of the detection pass (50 M events): packed / plain points x size hints / none.  GPU box: python tools/ab_pass.py"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import eventcalib_amd
from eventcalib_amd.pipeline import DetectPipeline
import synth_stream as SS
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
ctx = eventcalib_amd.Context(0)
ev = SS.make_stream(n, device="cuda")
t0, t1 = SS.tiled_windows(5.0, 5.0 + (n - 1) / 1e6)
for packed in (True, False):
    pipe = DetectPipeline(ctx, packed=packed)
    pipe.set_windows(t0, t1)
    pipe.set_detect_params(5, 36, 15.511363636363637)
    pipe.run(ev); torch.cuda.synchronize()
    S = len(t0)
    mw = int((pipe.win_hi[:S] - pipe.win_lo[:S]).max().item()); ms = int(pipe.seg_cnt[:2 * S].max().item())
    for hints in ((mw, ms), (0, 0)):
        for what in ("full", "slice", "noextract"):
            kw = dict(max_win_events=hints[0], max_seg_points=hints[1])
            if what == "slice": kw["slice_only"] = True
            if what == "noextract": kw["detect"] = False
            for _ in range(3): pipe.run(ev, **kw)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): pipe.run(ev, **kw)
            b.record(); torch.cuda.synchronize()
            print("packed=%d hints=%s %-9s %.3f ms" % (packed, hints != (0, 0), what, a.elapsed_time(b) / 10))
