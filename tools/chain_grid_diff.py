"""Which windows of the chain test's keyframe search get another verdict from the product's grid finder than from the ground-truth
rule of tests/oracle_chain.py?  GPU box: python tools/chain_grid_diff.py [events]; writes gpurun_out/chain_grid_diff.npz"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import eventcalib_amd
import eventcalib_amd.capi as capi
import synth_stream as SS
import oracle_chain as OC
import oracle_lib as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
SS.TRAJECTORY = "orbit"
buf = SS.make_stream(n, rate=1.0e6, t_start=5.0, device="cpu", seed=21)
ev = buf.cuda()
lm = SS.landmarks()
def centres_at(t):
    R, C = SS.pose(torch.tensor([t], dtype=torch.float64))
    return SS.project(lm, R.expand(36, 3, 3), C.expand(36, 3)).numpy()
ctx = eventcalib_amd.Context(0)
W = OC.WindowOracle(buf.numpy(), OC.gt_grid_order(centres_at))
diffs = []
detail = []
def detect(t0, t1):
    r = W.window(t0, t1)
    packed = capi.detect_pass(ctx, ev.data_ptr(), n, np.array([t0]), np.array([t1]), 65536, 4.0, 2, 5, 9, 4)
    found = (int(packed[0, 0]) & 0xFF) == 0 and packed[0, 1] != 0
    feat = packed[0, 3:].reshape(36, 3)
    same = found == r["found"] and (not found or np.array_equal(feat, r["features"]))
    if not same or int(packed[0, 2]) != r["events_num"]:
        lo, hi = O.window_bounds(W.rec, t0, t1)
        out = O.detect_windows_full(W.rec, [t0], [t1], [0, hi - lo], max(hi - lo, 1), 4.0, 2, 5, 36, OC.RADIUS_THR, n_threads=1)
        nc = int(out["win_info"][0][0])
        cand = out["cand_xyr"][:nc]
        gt = centres_at(0.5 * (t0 + t1))
        d = np.linalg.norm(cand[None, :, :2] - gt[:, None, :], axis=2)
        print("window [%.6f, %.6f] len %.1f steps: product found=%d oracle found=%d events %d/%d candidates %d; GT nearest distances max %.2f, sorted tail %s"
              % (t0, t1, (t1 - t0) / 5e-4, found, r["found"], int(packed[0, 2]), r["events_num"], nc, d.min(axis=1).max() if nc else -1,
                 np.round(np.sort(d.min(axis=1))[-4:], 2) if nc else None), flush=True)
        if found and r["found"]:
            bad = np.flatnonzero((feat != r["features"]).any(axis=1))
            print("   both found, model points that differ:", bad, feat[bad], r["features"][bad])
        diffs.append((t0, t1, found, r["found"], nc))
        if found and not r["found"] and len(detail) < 6:
            detail.append(dict(t0=t0, t1=t1, cand=cand.copy(), gt=gt.copy(), feat=feat.copy()))
    return r["found"], r["events_num"], r["features"]
kf = O.policy_run(detect, 5.0, 5.0 + (n - 1) / 1e6, 30, 5e-4, 4000, 9, 4, mode=1)
print("oracle-driven search: keyframes", len(kf["time"]), "windows", kf["windows"], "evaluated", W.calls, "differing", len(diffs))
np.savez(os.path.join(ROOT, "gpurun_out", "chain_grid_diff.npz"), diffs=np.array(diffs, float), **{"%s_%d" % (k, i): d[k] for i, d in enumerate(detail) for k in ("cand", "gt", "feat")})
