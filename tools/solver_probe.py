"""M2 alone: the benchmark's spline problem (45 M residuals, 2000 control points), ecal_solver_solve timed with its trace —
streamed evaluation (default) and ECAL_FORCE=solver_no_stream side by side.  `python tools/solver_probe.py [n_events] [iters] [reps]`"""
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
VAR = os.environ.get("PROBE_VARIANT", "")
if "threads" in VAR:
    torch.set_num_threads(16)
    a = torch.randn(2000, 2000); (a @ a).sum().item()
    np.polyfit(np.arange(3.0), np.arange(3.0), 1)
if "pipe" in VAR:
    import synth_stream as SS
    from eventcalib_amd.pipeline import DetectPipeline
    ev = SS.make_stream(50_000_000, device="cuda")
    pipe = DetectPipeline(ctx)
    t0w, t1w = SS.tiled_windows(5.0, 5.0 + (50_000_000 - 1) / 1e6)
    pipe.set_windows(t0w, t1w)
    for _ in range(3):
        pipe.run(ev)
    torch.cuda.synchronize()
for mode in ("stream", "plain", "stream", "plain")[:int(os.environ.get("PROBE_MODES", "4"))]:
    if mode == "plain":
        os.environ["ECAL_FORCE"] = "solver_no_stream"
    else:
        os.environ.pop("ECAL_FORCE", None)
    os.environ["ECAL_TRACE"] = "solver"
    ctx.reload_env()
    s = Solver(ctx, prob)
    opt = s.default_options()
    opt.max_num_iterations = 2
    s.solve(x0, opt)
    opt.max_num_iterations = iters
    for r in range(reps):
        if "evals" in VAR:
            d_x = torch.as_tensor(x0, device="cuda"); d_acc = torch.empty(s.n_normal, dtype=torch.float64, device="cuda")
            st = torch.cuda.current_stream()
            for _ in range(6):
                s.evaluate_dev(d_x.data_ptr(), 1, d_acc.data_ptr(), st.cuda_stream)
            for _ in range(5):
                s.evaluate_dev(d_x.data_ptr(), 0, d_acc.data_ptr(), st.cuda_stream)
        if "sleep" in VAR:
            time.sleep(0.3)
        torch.cuda.synchronize(); t = time.perf_counter()
        xs, summ = s.solve(x0, opt)
        torch.cuda.synchronize(); el = time.perf_counter() - t
        print("%s: %d iterations in %.4f s = %.1f it/s; successful %d, final cost %.9e" % (mode, summ.iterations, el, summ.iterations / el, summ.successful_steps, summ.final_cost), flush=True)
    s.close()
