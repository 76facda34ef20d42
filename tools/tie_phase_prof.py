"""Debug tool: shader cycles of the phases of resolve_ties_inline (extract_window.hpp) — builds a -DECAL_PHASE_PROF copy of the
library under /tmp; never used by tests or bench.  Run on the GPU box:  python tools/tie_phase_prof.py [events]"""
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
import eventcalib_amd
from eventcalib_amd.pipeline import DetectPipeline
import synth_stream as SS
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
ctx = eventcalib_amd.Context(0); pipe = DetectPipeline(ctx)
ev = SS.make_stream(n, device="cuda")
t0, t1 = SS.tiled_windows(5.0, 5.0 + (n - 1) / 1e6)
pipe.set_windows(t0, t1)
pipe.set_detect_params(5, 36, 15.511363636363637)
L = capi.load_library()
d = (ctypes.c_ulonglong * 16)()
pipe.run(ev); torch.cuda.synchronize()
L.ecal_debug_det_cycles(d, 1)
pipe.run(ev); torch.cuda.synchronize()
L.ecal_debug_det_cycles(d, 0)
d = list(d)
wg, ncl = max(d[8], 1), max(d[13], 1)
print("extract kernel, %d staged workgroups (cycles per workgroup): stage-in %.0f  label/scatter %.0f  rank scan + ties %.0f  pairing + circle test %.0f  write-out %.0f"
      % (wg, d[0] / wg, d[1] / wg, d[2] / wg, d[3] / wg, d[4] / wg))
print("ties resolved in the first pass: %d clusters of %.1f members on average; cycles per cluster: tree staging (per polarity with ties) %.0f | "
      "range queries %.0f | queue simulation %.0f | order + nth_element %.0f" % (d[13], d[14] / ncl, d[9] / ncl, d[10] / ncl, d[11] / ncl, d[12] / ncl))
