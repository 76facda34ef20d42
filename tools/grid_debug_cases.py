import sys, os
os.environ["ECAL_TRACE"] = "grid"
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import eventcalib_amd
import synth_stream as SS
import test_gpu_grid as TG
ctx = eventcalib_amd.Context(0)
for min_dist in (12.0, 8.0):
    rng = np.random.default_rng(int(min_dist) + 100)
    views = list(TG._project_centres(torch, np.linspace(5.0, 9.0, 30))) + TG._tilted_views(torch, [20, 30, 40, 45, 50] * 6, seed=9)
    cases = []
    for gt in views:
        if not ((gt[:, 0].min() > 0) and (gt[:, 0].max() < SS.SENSOR_W) and (gt[:, 1].min() > 0) and (gt[:, 1].max() < SS.SENSOR_H)): continue
        p = gt + rng.normal(0, 0.7, size=(36, 2))
        k = int(rng.integers(3, 16))
        step = np.sort(np.linalg.norm(p[:, None] - p[None], axis=2) + 1e9 * np.eye(36), axis=1)[:, 0].min()
        cl = TG._inside_hull_points(rng, p, k, p, min(min_dist, 0.4 * step), on_edge=k // 3)
        allp = np.concatenate([p, cl]); perm = rng.permutation(len(allp))
        cases.append(allp[perm])
    order, found = TG._run_grid(ctx, torch, cases)
    for s in range(len(cases)):
        if not found[s]:
            print(min_dist, "case", s, "n", len(cases[s]), "dbg", order[s][8:16])
            np.save("gpurun_out/grid_case_%d_%d.npy" % (int(min_dist), s), cases[s])
