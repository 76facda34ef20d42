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
# marks of dbscan_pixel.hpp (thread 0 of every workgroup, shader-clock cycles between the marks, summed over the launch's workgroups)
rows = [(7, "A   scalar loads: segment count + offset"), (9, "A   own points arrive (global -> registers)"),
        (12, "A+B.0  bbox, fits vote, wave 0 replays 64 inserts (incl. waiting for the slowest wave's points)"),
        (13, "B.1 walks of the top tree + first bids (2 barriers)"), (14, "B.2 level-synchronous bidding (1 barrier per level)"),
        (1, "bitmap clear + set (2 barriers)"), (5, "raster ranks: row prefix, block scan, rank table (4 barriers)"),
        (2, "D   core test (1 barrier)"), (6, "E.1 union-find over the half disc (1 barrier)"),
        (3, "E.2/3 flatten + one-way edges"), (4, "F   seed ranks + labels out (2 barriers)")]
tot = sum(v[i] for i, _ in rows) + v[0]
print("dbscan_pixel_kernel<16, 768>: cycles per workgroup (= per segment), %d workgroups" % wg)
for i, nm in rows:
    print("  %-100s %9.0f  %5.1f %%" % (nm, v[i] / wg, 100.0 * v[i] / tot))
print("  total %.0f cycles/WG; tree levels below the top tree %.1f per WG; one-way edges %.3f per WG" % (tot / wg, v[8] / wg, v[11] / wg))
print("  barriers per workgroup: 17 + levels = %.0f; longest dependent chain: B.2 (per level: LDS read -> compare -> ds_min -> barrier)" % (17 + v[8] / wg))
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
