"""Debug: the device policy's look-ahead (ecal_detect_keyframes follows every piece's likely chain of windows) against the
same call with one window per piece and pass (ECAL_ADAPTIVE_DEPTH=1: the reference's loop as it stands, which the tests pin
on the policy oracle) over random streams, rates and piece counts: same keyframes, same windows."""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # (tests/ holds the oracle-checked fuzzers: only tests may call the oracle)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import eventcalib_amd
from eventcalib_amd.adaptive import detect_keyframes_device
import synth_stream as SS
ctx = eventcalib_amd.Context(0)
n_ok = 0
for seed, rate, pieces in itertools.product(range(int(sys.argv[1]) if len(sys.argv) > 1 else 3), (0.7e6, 1.0e6, 2.0e6), (1, 5, 37, 300)):
    n = 1_200_000
    ev = SS.make_stream(n, rate=rate, device="cuda", seed=500 + seed, noise_frac=0.05 + 0.05 * seed)
    torch.cuda.synchronize()
    t_first, t_last = 5.0, 5.0 + (n - 1) / rate
    os.environ.pop("ECAL_ADAPTIVE_DEPTH", None); os.environ.pop("ECAL_ADAPTIVE_DEPTH_MAX", None)
    a = detect_keyframes_device(ctx, ev, 5e-4, 4000, pieces, t_first, t_last)
    os.environ["ECAL_ADAPTIVE_DEPTH"] = "1"; os.environ["ECAL_ADAPTIVE_DEPTH_MAX"] = "1"
    b = detect_keyframes_device(ctx, ev, 5e-4, 4000, pieces, t_first, t_last)
    for k in ("time", "duration", "events_num", "features"):
        assert np.array_equal(a[k], b[k]), (seed, rate, pieces, k)
    assert a["steps"] == b["steps"] and a["windows"] == b["windows"], (seed, rate, pieces)
    n_ok += 1
    print("seed %d rate %.1f pieces %d: %d keyframes, %d windows, longest chain %d" % (seed, rate / 1e6, pieces, len(a["time"]), a["windows"], a["steps"]), flush=True)
print("all", n_ok, "runs: look-ahead == one window per pass")
