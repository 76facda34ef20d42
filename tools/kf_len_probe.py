import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, eventcalib_amd, synth_stream as SS
from eventcalib_amd.adaptive import detect_keyframes_device
import eventcalib_amd.capi as capi
n = 50_000_000
ctx = eventcalib_amd.Context(0)
ev = SS.make_stream(n, device="cuda")
kf = detect_keyframes_device(ctx, ev, 5e-4, 4000, 1270, 5.0, 5.0 + (n - 1) / 1e6, gate_mode=capi.GATE_SHARED_MAP)
d = kf["duration"]
L = np.rint((d[:, 1] - d[:, 0]) / 5e-4).astype(int)
print("keyframes", len(L), "window length (steps) histogram:", {int(k): int(v) for k, v in zip(*np.unique(L, return_counts=True))})
t = np.sort(kf["time"])
gap = np.rint(np.diff(t) / 5e-4).astype(int)
u, c = np.unique(np.minimum(gap, 40), return_counts=True)
print("steps between consecutive keyframes (40 = more):", {int(k): int(v) for k, v in zip(u, c)})
print("events per keyframe window: min %d median %d max %d" % (kf["events"].min(), np.median(kf["events"]), kf["events"].max()) if "events" in kf else kf.keys())
