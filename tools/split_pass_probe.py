"""Probe: the pass of the hot path with its windows cut into K parts, each part's stage chain on its own stream and context —
the drain of one part's kernel is filled by the other parts' workgroups.  `python tools/split_pass_probe.py [n_events] [K...]`"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import eventcalib_amd, synth_stream as SS
from eventcalib_amd.pipeline import DetectPipeline
from eventcalib_amd.capi import PackedPoints
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
Ks = [int(a) for a in sys.argv[2:]] or [1, 2, 3, 4]
dev = torch.device("cuda", 0)
ev = SS.make_stream(n, device="cuda")
ctx0 = eventcalib_amd.Context(0)
pipe = DetectPipeline(ctx0, want_event_point=False)
t0, t1 = SS.tiled_windows(5.0, 5.0 + (n - 1) / 1e6)
pipe.set_windows(t0, t1)
pipe.set_detect_params(5, 36, 15.511363636363637)
pipe.run(ev)
torch.cuda.synchronize()
S = pipe.S
ref = {k: getattr(pipe, k).clone() for k in ("labels", "win_info", "cand_pair", "cand_xyr", "kept_labels", "rep", "n_clusters")}
eps, minpts = 4.0, 2
ctxs = [ctx0] + [eventcalib_amd.Context(0) for _ in range(max(Ks) - 1)]
streams = [torch.cuda.Stream(dev) for _ in range(max(Ks))]

def split_pass(K):
    main = streams[0]
    c0 = ctxs[0]
    c0.window_bounds_dev(ev.data_ptr(), n, pipe.t0.data_ptr(), pipe.t1.data_ptr(), S, pipe.win_lo.data_ptr(), pipe.win_hi.data_ptr(),
                         pipe.win_base.data_ptr(), main.cuda_stream)
    e0 = torch.cuda.Event(); e0.record(main)
    cuts = [S * k // K for k in range(K + 1)]
    for k in range(K):
        a, b = cuts[k], cuts[k + 1]
        st = streams[k]
        if k:
            st.wait_event(e0)
        c = ctxs[k]
        Sw = b - a
        i4 = 4
        pk = PackedPoints(pipe.xy16.data_ptr(), pipe.seg_fmt.data_ptr() + 2 * a * i4)
        c.slice_events_packed_dev(ev.data_ptr(), n, pipe.win_lo.data_ptr() + a * i4, pipe.win_hi.data_ptr() + a * i4, pipe.win_base.data_ptr() + a * i4,
                                  Sw, 0, n, pipe._xy.data_ptr(), pipe.seg_off.data_ptr() + 2 * a * i4, pipe.seg_cnt.data_ptr() + 2 * a * i4, 0,
                                  pipe.flags.data_ptr(), pk, st.cuda_stream)
        c.dbscan_batch_packed_dev(pipe._xy.data_ptr(), pipe.seg_off.data_ptr() + 2 * a * i4, pipe.seg_cnt.data_ptr() + 2 * a * i4, 2 * Sw, n, 0, eps, minpts,
                                  pipe.labels.data_ptr(), pipe.n_clusters.data_ptr() + 2 * a * i4, pk, st.cuda_stream)
        c.extract_batch_packed_dev(pipe._xy.data_ptr(), pipe.seg_off.data_ptr() + 2 * a * i4, pipe.seg_cnt.data_ptr() + 2 * a * i4, pipe.labels.data_ptr(),
                                   pipe.n_clusters.data_ptr() + 2 * a * i4, Sw, n, eps, pipe.det[0], pipe.det[1], pipe.det[2],
                                   pipe.win_info.data_ptr() + 4 * a * i4, pipe.cand_pair.data_ptr(), pipe.cand_xyr.data_ptr(),
                                   pipe.kept_labels.data_ptr(), pipe.rep.data_ptr(), pk, st.cuda_stream, fit_circle=pipe.det[3], knn_num=pipe.det[4])
        if k:
            e = torch.cuda.Event(); e.record(st); main.wait_event(e)

for K in Ks:
    for _ in range(3):
        split_pass(K)
    torch.cuda.synchronize()
    for k in ref:
        getattr(pipe, k).zero_() if k not in ("n_clusters",) else None
    split_pass(K)
    torch.cuda.synchronize()
    same = all(torch.equal(getattr(pipe, k)[:ref[k].shape[0]], ref[k]) for k in ("win_info", "n_clusters"))
    npts = int(pipe.seg_off[2 * S - 1].item() + pipe.seg_cnt[2 * S - 1].item())
    same = same and all(torch.equal(getattr(pipe, k)[:npts], ref[k][:npts]) for k in ("labels", "kept_labels", "rep"))
    reps = 20
    t = time.perf_counter()
    for _ in range(reps):
        split_pass(K)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t) / reps
    print("K = %d: %.4f ms per pass, %.0f Mevents/s, same results: %s" % (K, el * 1e3, n / el / 1e6, same), flush=True)
