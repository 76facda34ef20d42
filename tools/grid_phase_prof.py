"""Debug tool: per-phase shader cycles of grid_order_kernel over one adaptive-window search (builds a -DECAL_PHASE_PROF copy of the
library under /tmp; never used by tests or bench).  Run on the GPU box:  python tools/grid_phase_prof.py"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
src = os.path.join(ROOT, "eventcalib_amd", "csrc")
out = "/tmp/libecal_prof.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                       "-DECAL_PHASE_PROF", "-shared", "-o", out] + ["-L/opt/rocm/lib", "-lrccl"] + sorted(os.path.join(src, f) for f in os.listdir(src) if f.endswith(".hip")))
import numpy as np, torch
import eventcalib_amd.capi as capi
capi.lib_path = lambda: out
import eventcalib_amd, synth_stream as SS
from eventcalib_amd.adaptive import detect_keyframes_device
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
pieces = int(sys.argv[2]) if len(sys.argv) > 2 else 1270
ctx = eventcalib_amd.Context(0)
ev = SS.make_stream(n, device="cuda")
L = capi.load_library()
b = (ctypes.c_ulonglong * 16)()
detect_keyframes_device(ctx, ev, 5e-4, 4000, pieces, 5.0, 5.0 + (n - 1) / 1e6)
L.ecal_debug_grid_cycles(b, 1)
L.ecal_debug_grid_cycles(b, 3)
L.ecal_debug_grid_cycles(b, 4)
kf = detect_keyframes_device(ctx, ev, 5e-4, 4000, pieces, 5.0, 5.0 + (n - 1) / 1e6)
L.ecal_debug_grid_cycles(b, 0)
v = list(b); w = max(v[15], 1); tot = sum(v[:7])
names = ["seed + basis", "first walk", "first match", "restart / loop", "homography fits", "sweeps' searches", "matches after sweeps"]
print("%d windows with >= M candidates, %d keyframes; cycles per such window: %.0f" % (w, len(kf["time"]), tot / w))
for i, nm in enumerate(names):
    print("  %-22s %9.0f  %5.1f %%" % (nm, v[i] / w, 100.0 * v[i] / tot))
h = (ctypes.c_ulonglong * 16)()
L.ecal_debug_grid_cycles(h, 2)
h = list(h)
print("windows by cycles: " + "  ".join("2^%d: %d" % (i + 10, h[i]) for i in range(14) if h[i]))
mx = h[14]
print("slowest window: %d cycles, %d candidates, %d nodes at the end, %d sweeps" % (mx >> 24, (mx >> 16) & 255, (mx >> 8) & 255, mx & 255))

o = (ctypes.c_ulonglong * 16)()
L.ecal_debug_grid_cycles(o, 4)
o = list(o)
for i, nm in enumerate(["found: first start, first walk", "found: first start, second attempt", "found: later start, first walk", "found: later start, second attempt",
                        "not found, more candidates than pattern points", "not found, exactly as many"]):
    if o[i]:
        print("  %-50s %6d windows, %8.0f cycles each, %5.1f %% of all cycles" % (nm, o[i], o[8 + i] / o[i], 100.0 * o[8 + i] / max(1, sum(o[8:14]))))
