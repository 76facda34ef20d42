"""M2 alone: the benchmark's spline problem (45 M residuals, 2000 control points), ecal_solver_solve timed with its trace —
streamed evaluation (default) and ECAL_SOLVER_NO_STREAM=1 side by side.  `python tools/solver_probe.py [n_events] [iters] [reps]`"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
T = str(len(os.sched_getaffinity(0)))
for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(k, "16")
import numpy as np, torch
import eventcalib_amd
from eventcalib_amd.capi import Solver
import synth_solver_torch as ST
n_events = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
duration = n_events / 1e6
n_cp = max(4, int(duration / (50 * 5e-4)))
n_res = int(0.9 * n_events)
ctx = eventcalib_amd.Context(0)
prob, x = ST.make_problem(n_res, n_cp, 5.0, 5.0 + duration, seed=777, device="cuda", round_pixels=True)
rng = np.random.default_rng(99)
x0 = x.copy()
x0[:4] *= 1 + 0.01 * rng.uniform(-1, 1, 4)
x0[4:9] += 0.01 * rng.uniform(-1, 1, 5)
for mode in ("stream", "plain", "stream", "plain"):
    if mode == "plain":
        os.environ["ECAL_SOLVER_NO_STREAM"] = "1"
    else:
        os.environ.pop("ECAL_SOLVER_NO_STREAM", None)
    os.environ["ECAL_SOLVER_TRACE"] = "1" if os.environ.get("TRACE", "1") == "1" else "0"
    ctx.reload_env()
    s = Solver(ctx, prob)
    opt = s.default_options()
    opt.max_num_iterations = 2
    s.solve(x0, opt)
    opt.max_num_iterations = iters
    for r in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter()
        xs, summ = s.solve(x0, opt)
        torch.cuda.synchronize(); el = time.perf_counter() - t
        print("%s: %d iterations in %.4f s = %.1f it/s; successful %d, final cost %.9e" % (mode, summ.iterations, el, summ.iterations / el, summ.successful_steps, summ.final_cost), flush=True)
    s.close()
