"""Soak of the streamed LM solve: many solves back to back on problems of several sizes (partitions of 5 - 16 interiors, groups with
and without chunks), each checked against the plain form's final cost; prints the slowest and the mean solve per problem.
`python tools/solver_soak.py [rounds]`"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import eventcalib_amd
from eventcalib_amd.capi import Solver
import synth_solver as SV
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ctx = eventcalib_amd.Context(0)
probs = []
for (n_cp, n_res, noise) in ((700, 200000, 0.3), (520, 60000, 0.0), (1500, 400000, 0.5), (2000, 30000, 0.2)):
    rng = np.random.default_rng(n_cp)
    prob, x_gt = SV.make_problem(n_res, n_cp=n_cp, seed=n_cp, pixel_noise=noise)
    x0 = SV.perturb(x_gt, n_cp, rng)
    os.environ["ECAL_FORCE"] = "solver_no_stream"; ctx.reload_env()
    s = Solver(ctx, prob); opt = s.default_options(); opt.max_num_iterations = 10
    xp, sp = s.solve(x0, opt); s.close()
    os.environ.pop("ECAL_FORCE", None); ctx.reload_env()
    probs.append((prob, x0, sp.final_cost, sp.iterations, Solver(ctx, prob), sp.initial_cost))
worst = [0.0] * len(probs); tot = [0.0] * len(probs); n = [0] * len(probs)
for r in range(rounds):
    for i, (prob, x0, want, iters, s, init) in enumerate(probs):
        opt = s.default_options(); opt.max_num_iterations = 10
        t = time.perf_counter()
        x, sm = s.solve(x0, opt)
        el = time.perf_counter() - t
        # (a noise-free problem ends near zero: the difference is measured against the initial cost too)
        assert sm.iterations == iters and abs(sm.final_cost - want) <= 1e-9 * want + 1e-12 * init, (r, i, sm.iterations, iters, sm.final_cost, want)
        worst[i] = max(worst[i], el); tot[i] += el; n[i] += 1
for i in range(len(probs)):
    print("problem %d: %d solves, mean %.2f ms, slowest %.2f ms" % (i, n[i], 1e3 * tot[i] / n[i], 1e3 * worst[i]), flush=True)
print("all", sum(n), "streamed solves == the plain solve")
