"""The device policy's look-ahead (ecal_detect_keyframes follows every piece's likely chain of windows) against the
same call with one window per piece and pass (ECAL_ADAPTIVE_SHAPE=depth=1,depth_max=1: the reference's loop as it stands, which the tests pin
on the policy oracle) over random streams, rates and piece counts: same keyframes, same windows.
`python tests/fuzz_policy.py N` runs N seeds; tests/test_gpu_fuzz.py runs a bounded, fixed-seed sweep as a -m gpu test."""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # (tests/ holds the oracle-checked fuzzers: only tests may call the oracle)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

RATES = (0.7e6, 1.0e6, 2.0e6)
PIECES = (1, 5, 37, 300)


def run(seeds, rates=RATES, pieces_list=PIECES, ctx=None, verbose=True, n=1_200_000):
    import torch
    import eventcalib_amd
    from eventcalib_amd.adaptive import detect_keyframes_device
    import synth_stream as SS
    own = ctx is None
    if own:
        ctx = eventcalib_amd.Context(0)
    n_ok = n_kf = 0
    saved = {k: os.environ.get(k) for k in ("ECAL_ADAPTIVE_SHAPE",)}
    try:
        for seed, rate, pieces in itertools.product(seeds, rates, pieces_list):
            ev = SS.make_stream(n, rate=rate, device="cuda", seed=500 + seed, noise_frac=0.05 + 0.05 * (seed % 6))
            torch.cuda.synchronize()
            t_first, t_last = 5.0, 5.0 + (n - 1) / rate
            os.environ.pop("ECAL_ADAPTIVE_SHAPE", None)
            __import__("eventcalib_amd.capi", fromlist=["sync_env"]).sync_env()   # (the switches are read once per context)
            a = detect_keyframes_device(ctx, ev, 5e-4, 4000, pieces, t_first, t_last)
            os.environ["ECAL_ADAPTIVE_SHAPE"] = "depth=1,depth_max=1"
            __import__("eventcalib_amd.capi", fromlist=["sync_env"]).sync_env()   # (the switches are read once per context)
            b = detect_keyframes_device(ctx, ev, 5e-4, 4000, pieces, t_first, t_last)
            for k in ("time", "duration", "events_num", "features"):
                assert np.array_equal(a[k], b[k]), (seed, rate, pieces, k)
            assert a["steps"] == b["steps"] and a["windows"] == b["windows"], (seed, rate, pieces)
            n_ok += 1
            n_kf += len(a["time"])
            if verbose:
                print("seed %d rate %.1f pieces %d: %d keyframes, %d windows, longest chain %d" % (seed, rate / 1e6, pieces, len(a["time"]), a["windows"], a["steps"]), flush=True)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
                __import__("eventcalib_amd.capi", fromlist=["sync_env"]).sync_env()   # (the switches are read once per context)
            else:
                os.environ[k] = v
                __import__("eventcalib_amd.capi", fromlist=["sync_env"]).sync_env()   # (the switches are read once per context)
        if own:
            ctx.close()
    return dict(runs=n_ok, keyframes=n_kf)


if __name__ == "__main__":
    r = run(range(int(sys.argv[1]) if len(sys.argv) > 1 else 3))
    print("all", r["runs"], "runs: look-ahead == one window per pass")
