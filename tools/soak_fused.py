"""Debug: the fused detection pass against the three stage calls, many times over large streams — a stale read of a value the
workgroup wrote a moment earlier (scalar cache, vector L1) would show up as a rare mismatch here, never in a single small run."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import eventcalib_amd
from eventcalib_amd.pipeline import DetectPipeline
import synth_stream as SS
import test_gpu_fused as TF
ctx = eventcalib_amd.Context(0)
env = (ctx, DetectPipeline, torch)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8_000_000
for rep in range(reps):
    rate = (1.0e6, 0.7e6, 1.3e6)[rep % 3]
    ev = SS.make_stream(n, rate=rate, device="cuda", seed=500 + rep)
    t0, t1 = SS.tiled_windows(5.0, 5.0 + (n - 1) / rate)
    ref = TF._both(env, ev, t0, t1)
    print("rep %d rate %.1f: %d windows, %d points identical in both forms" % (rep, rate / 1e6, len(t0), int(ref["seg_cnt"].sum())), flush=True)
print("all equal")
