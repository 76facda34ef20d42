"""The normal-equations launch alone (memsets + normal_eq_kernel + head reduction) on the benchmark's spline problem."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import eventcalib_amd
from eventcalib_amd.capi import Solver
import synth_solver_torch as ST
n_events = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
duration = n_events / 1e6
n_cp = max(4, int(duration / (50 * 5e-4)))
ctx = eventcalib_amd.Context(0)
prob, x = ST.make_problem(int(0.9 * n_events), n_cp, 5.0, 5.0 + duration, seed=777, device="cuda", round_pixels=True)
s = Solver(ctx, prob)
st = torch.cuda.current_stream()
d_x = torch.as_tensor(x, device="cuda"); d_acc = torch.empty(s.n_normal, dtype=torch.float64, device="cuda")
for rnd in range(3):
    for _ in range(3):
        s.evaluate_dev(d_x.data_ptr(), 1, d_acc.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(10):
        s.evaluate_dev(d_x.data_ptr(), 1, d_acc.data_ptr(), st.cuda_stream)
    e1.record(st)
    torch.cuda.synchronize()
    print("one role", "%.4f ms per evaluation" % (e0.elapsed_time(e1) / 10), flush=True)
