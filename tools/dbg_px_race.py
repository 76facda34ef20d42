"""Debug: pixel DBSCAN kernel vs general tiers on the benchmark stream, several runs; prints differing segments."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
import eventcalib_amd
from eventcalib_amd.pipeline import DetectPipeline
import synth_stream as SS
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
rate = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0e6
wlen = float(sys.argv[3]) if len(sys.argv) > 3 else 1.5e-3
ctx = eventcalib_amd.Context(0); pipe = DetectPipeline(ctx)
ev = SS.make_stream(n, rate=rate, device="cuda")
t0, t1 = SS.tiled_windows(5.0, 5.0 + (n - 1) / rate, wlen)
pipe.set_windows(t0, t1)
S = len(t0)
os.environ["ECAL_FORCE"] = "dbscan_general"
pipe.run(ev, detect=False); torch.cuda.synchronize()
ref_l = pipe.labels.clone(); ref_n = pipe.n_clusters.clone()
os.environ.pop("ECAL_FORCE", None)
off = pipe.seg_off[:2 * S].long().cpu().numpy(); cnt = pipe.seg_cnt[:2 * S].long().cpu().numpy()
o_t, c_t = pipe.seg_off[:2 * S].long(), pipe.seg_cnt[:2 * S].long()
used = torch.repeat_interleave(o_t, c_t) + (torch.arange(int(c_t.sum()), device="cuda") - torch.repeat_interleave(torch.cumsum(c_t, 0) - c_t, c_t))
for rep in range(12):
    pipe.labels.fill_(-7)
    pipe.run(ev, detect=False); torch.cuda.synchronize()
    dn = (pipe.n_clusters[:2 * S] != ref_n[:2 * S]).nonzero().flatten().cpu().numpy()
    dl = used[(pipe.labels[used] != ref_l[used])].sort().values.cpu().numpy()
    print("run", rep, "segments with different n_clusters:", len(dn), "slots with different labels:", len(dl))
    if len(dl):
        seg = np.searchsorted(off, dl[0], side="right") - 1
        a, c = off[seg], cnt[seg]
        print("  first bad segment", seg, "n", c, "ncl", int(pipe.n_clusters[seg]), "ref", int(ref_n[seg]))
        got = pipe.labels[a:a + c].cpu().numpy(); want = ref_l[a:a + c].cpu().numpy()
        bad = np.nonzero(got != want)[0]
        print("  bad pids", bad[:20], "got", got[bad[:20]], "want", want[bad[:20]])
        np.save(os.path.join(ROOT, "gpurun_out", "bad_seg_%d.npy" % rep), pipe.xy[a:a + c].cpu().numpy())
