"""Debug tool: runs bounds + slicing + DBSCAN only (no extraction) on the benchmark stream with the library named by
ECAL_LIB (a -DECAL_PX_STOP=k build leaves the pixel DBSCAN kernel after phase k); meant to run under
`rocprofv3 --pmc ...` (tools/px_stop_probe.sh).  Never used by tests or bench."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import eventcalib_amd.capi as capi
if os.environ.get("ECAL_LIB"):
    capi.lib_path = lambda: os.environ["ECAL_LIB"]
import eventcalib_amd
from eventcalib_amd.pipeline import DetectPipeline
import synth_stream as SS
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
ctx = eventcalib_amd.Context(0); pipe = DetectPipeline(ctx)
ev = SS.make_stream(n, device="cuda")
t0, t1 = SS.tiled_windows(5.0, 5.0 + (n - 1) / 1e6)
pipe.set_windows(t0, t1)
for _ in range(3):
    pipe.run(ev, detect=bool(os.environ.get("ECAL_PROBE_DETECT")), slice_only=bool(os.environ.get("ECAL_PROBE_SLICE_ONLY")))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
print("ok")
