import os, sys, cProfile, pstats
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, eventcalib_amd, synth_stream as SS
from eventcalib_amd.calibrate import calibrate_stream
n = 50_000_000
SS.TRAJECTORY = "orbit"
ev = SS.make_stream(n, rate=1e6, t_start=5.0, device="cuda", seed=21)
SS.TRAJECTORY = "hover"
ctx = eventcalib_amd.Context(0)
calibrate_stream(ctx, ev, 5.0, 5.0 + (n - 1) / 1e6, piece_num=1270)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
r = calibrate_stream(ctx, ev, 5.0, 5.0 + (n - 1) / 1e6, piece_num=1270)
pr.disable()
print(r["stage_seconds"])
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
