import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import eventcalib_amd
from eventcalib_amd.pipeline import DetectPipeline
import synth_stream as SS
import test_gpu_grid as TG
ctx = eventcalib_amd.Context(0)
for rate, noise in ((4e6, 0.15), (4e6, 0.3), (4e6, 0.5), (6e6, 0.4), (8e6, 0.5)):
    pipe = DetectPipeline(ctx)
    n = 1_500_000
    buf = SS.make_stream(n, rate=rate, device="cpu", seed=8, noise_frac=noise)
    t, _, _ = SS.unpack_records(buf)
    t0, t1 = SS.tiled_windows(float(t[0]), float(t[-1]))
    pipe.set_windows(t0, t1)
    pipe.run(buf.cuda())
    order, found = pipe.order_grid(9, 4)
    torch.cuda.synchronize()
    S = len(t0)
    info = pipe.win_info[:S].cpu().numpy(); found = found.cpu().numpy()
    off = pipe.seg_off[:2 * S].cpu().numpy(); xyr = pipe.cand_xyr.cpu().numpy()
    gt = TG._project_centres(torch, (np.asarray(t0) + np.asarray(t1)) / 2)
    ok = info[:, 3] == 0
    gt36 = more = comp = compmore = 0
    for s in range(S):
        if not ok[s]: continue
        c = xyr[off[2 * s]: off[2 * s] + info[s, 0], :2]
        d = np.linalg.norm(c[:, None] - gt[s][None], axis=2) if len(c) else np.zeros((0, 36))
        complete = len(c) >= 36 and (d.min(axis=0) < 14.0).all()
        comp += complete; more += info[s, 0] > 36; compmore += complete and info[s, 0] > 36
    print(rate, noise, "windows", S, "paired", int(ok.sum()), "n>36", more, "complete", comp, "complete&n>36", compmore, "found", int(found.sum()), "max n", int(info[ok, 0].max() if ok.any() else 0))
