"""Debug tool: per-phase shader cycles of the DBSCAN kernel (builds a -DECAL_PHASE_PROF copy of the
library under /tmp; never used by tests or bench).  Run on the GPU box:  python tools/phase_prof.py"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
src = os.path.join(ROOT, "eventcalib_amd", "csrc")
out = "/tmp/libecal_prof.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                       "-DECAL_PHASE_PROF", "-shared", "-o", out] + os.environ.get("ECAL_PROF_FLAGS", "").split() + ["-L/opt/rocm/lib", "-lrccl"] + sorted(os.path.join(src, f) for f in os.listdir(src) if f.endswith(".hip")))
import numpy as np, torch
import eventcalib_amd.capi as capi
capi.lib_path = lambda: out
import eventcalib_amd
from eventcalib_amd.pipeline import DetectPipeline
import synth_stream as SS
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
ctx = eventcalib_amd.Context(0); pipe = DetectPipeline(ctx)
ev = SS.make_stream(n, device="cuda")
t0, t1 = SS.tiled_windows(5.0, 5.0 + (n - 1) / 1e6)
pipe.set_windows(t0, t1)
L = capi.load_library()
buf = (ctypes.c_ulonglong * 16)()
pipe.run(ev); torch.cuda.synchronize()
L.ecal_debug_phase_cycles(buf, 1)
pipe.run(ev, max_win_events=2000, max_seg_points=1000); torch.cuda.synchronize()
L.ecal_debug_phase_cycles(buf, 1)
v = list(buf); wg = max(v[10], 1)
names = ["B kd-bounds", "C cell sort / bitmap build", "D count", "E labels (grid E.1 + E.2/3)", "F rank", "bitmap ranks", "bitmap E.1"]
tot = sum(v[:7]) + v[12] + v[13] + v[14]
for i, nm in enumerate(names):
    print("%-28s %10.0f cycles/WG  %5.1f %%" % (nm, v[i] / wg, 100.0 * v[i] / tot))
print("B split: init+B.0 %.0f  B.1 %.0f  B.2 %.0f  (rest = anc/flags)" % (v[12]/wg, v[13]/wg, v[14]/wg))
print("D: candidate visits/WG %.0f (per point %.1f)   wave-steps/WG %.0f (x64 = %.0f lane slots)" % (v[12]/wg, v[12]/wg/578.0, v[13]/wg, 64*v[13]/wg))
print("levels/WG %.1f  sweeps/WG %.2f  WGs %d  total cycles/WG %.0f" % (v[8] / wg, v[9] / wg, wg, tot / wg))
print("pixel kernel: count/offset loads %.0f  points arrive %.0f  bbox + first barriers %.0f   (then init+B.0 %.0f)" % (v[7] / wg, v[9] / wg, v[15] / wg, v[12] / wg))
d = (ctypes.c_ulonglong * 16)()
L.ecal_debug_det_cycles(d, 0)
d = list(d); dw = max(d[8], 1) / 2      # two timed runs accumulated? (reset only DBSCAN's) -> per call counts included
print("extract kernel (cycles/WG over %d staged WGs): stage-in %.0f  label/scatter %.0f  rank scan %.0f  pairing+circle test %.0f  write-out %.0f"
      % (d[8], d[0] / max(d[8], 1), d[1] / max(d[8], 1), d[2] / max(d[8], 1), d[3] / max(d[8], 1), d[4] / max(d[8], 1)))
b = (ctypes.c_ulonglong * 16)()
L.ecal_debug_bo_cycles(b, 1)
pipe.run(ev); torch.cuda.synchronize()
L.ecal_debug_bo_cycles(b, 0)
b = list(b); S2 = 2 * len(t0)
print("member-order kernel (cycles per segment, all tiers summed, %d segments): stage-in + marks %.0f  tie vote %.0f  untied exit %.0f  tree %.0f  range queries %.0f  queue simulation + order out %.0f"
      % (S2, b[0] / S2, b[1] / S2, b[2] / S2, b[3] / S2, b[4] / S2, b[5] / S2))
