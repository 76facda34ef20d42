"""ecal_detect_keyframes under the shared-map gate (speculation + verification rounds) against the sequential
single-worker oracle (oracle/policy_oracle.cpp mode 1) over random streams, rates, trajectories, piece counts — and the
own-piece gate against mode 0 on the same windows.  extractFeatures() of a window comes to the oracle from the product's
detection stages (ecal_detect_pass, cached): what is compared is the policy.
`python tests/fuzz_shared_map.py N` runs N seeds; tests/test_gpu_fuzz.py runs a bounded, fixed-seed sweep as a -m gpu test."""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # (tests/ holds the oracle-checked fuzzers: only tests may call the oracle)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

CASES = ((0.7e6, "hover"), (1.0e6, "orbit"), (2.0e6, "hover"), (1.4e6, "orbit"))
PIECES = (2, 9, 61, 333)


def run(seeds, cases=CASES, pieces_list=PIECES, ctx=None, verbose=True, n=1_500_000):
    import torch
    import eventcalib_amd
    import eventcalib_amd.capi as capi
    from eventcalib_amd.adaptive import detect_keyframes_device
    import synth_stream as SS
    import oracle_lib as O
    import test_gpu_adaptive as TA
    own = ctx is None
    if own:
        ctx = eventcalib_amd.Context(0)
    n_ok = n_kf = 0
    try:
        for seed, (rate, traj) in itertools.product(seeds, cases):
            SS.TRAJECTORY = traj
            try:
                ev = SS.make_stream(n, rate=rate, device="cuda", seed=900 + seed, noise_frac=0.05 + 0.04 * (seed % 6))
            finally:
                SS.TRAJECTORY = "hover"
            torch.cuda.synchronize()
            t_first, t_last = 5.0, 5.0 + (n - 1) / rate
            cache = {}
            detect = TA._oracle_detect(ctx, ev, n, cache)        # one cache per stream: the piece counts share its windows
            for pieces in pieces_list:
                ref = O.policy_run(detect, t_first, t_last, pieces, 5e-4, 4000, 9, 4, mode=1)
                own_ref = O.policy_run(detect, t_first, t_last, pieces, 5e-4, 4000, 9, 4, mode=0)
                dev = detect_keyframes_device(ctx, ev, 5e-4, 4000, pieces, t_first, t_last, gate_mode=capi.GATE_SHARED_MAP)
                TA._same_keyframes(dev, ref)
                TA._same_keyframes(detect_keyframes_device(ctx, ev, 5e-4, 4000, pieces, t_first, t_last, gate_mode=capi.GATE_OWN_PIECE), own_ref)
                n_ok += 1
                n_kf += len(ref["time"])
                if verbose:
                    print("seed %d rate %.1f %s pieces %d: shared map %d keyframes, own piece %d, %d windows" %
                          (seed, rate / 1e6, traj, pieces, len(ref["time"]), len(own_ref["time"]), ref["windows"]), flush=True)
    finally:
        if own:
            ctx.close()
    return dict(runs=n_ok, keyframes=n_kf)


if __name__ == "__main__":
    r = run(range(int(sys.argv[1]) if len(sys.argv) > 1 else 2))
    print("all", r["runs"], "runs: both gates == the policy oracle")
