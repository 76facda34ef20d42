"""Debug: the hash-table slicer against the general slicer on the benchmark stream, several runs (CAS / ds_min races would
show as differing outputs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import eventcalib_amd
from eventcalib_amd.pipeline import DetectPipeline
import synth_stream as SS
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
ctx = eventcalib_amd.Context(0); pipe = DetectPipeline(ctx)
ev = SS.make_stream(n, device="cuda")
t0, t1 = SS.tiled_windows(5.0, 5.0 + (n - 1) / 1e6)
pipe.set_windows(t0, t1)
S = len(t0)
os.environ["ECAL_FORCE"] = "slice_general"
pipe.run(ev, slice_only=True); torch.cuda.synchronize()
os.environ.pop("ECAL_FORCE", None)
ref = [x.clone() for x in (pipe.seg_off[:2 * S], pipe.seg_cnt[:2 * S], pipe.event_point[:n], pipe.xy[:n])]
used = torch.zeros(n, dtype=torch.bool, device="cuda")
o, c = ref[0].long(), ref[1].long()
idx = torch.repeat_interleave(o, c) + (torch.arange(int(c.sum()), device="cuda") - torch.repeat_interleave(torch.cumsum(c, 0) - c, c))
used[idx] = True
for rep in range(8):
    pipe.xy.fill_(-1.0); pipe.event_point.fill_(-5)
    pipe.run(ev, slice_only=True); torch.cuda.synchronize()
    ok = (torch.equal(pipe.seg_off[:2 * S], ref[0]) and torch.equal(pipe.seg_cnt[:2 * S], ref[1]) and
          torch.equal(pipe.event_point[:n], ref[2]) and torch.equal(pipe.xy[:n][used], ref[3][used]))
    print("run", rep, "identical to the general slicer:", ok)
