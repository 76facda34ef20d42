"""The whole detection pass (slicing, DBSCAN labels, extraction) against the CPU oracle on many small random streams:
event rates from 0.5 to 3.5 Mev/s (first passes, second passes and general tiers all get their share), noise 5-30 %.
`python tests/fuzz_pipeline.py N` runs N seeds (12 streams each); tests/test_gpu_fuzz.py runs a bounded, fixed-seed sweep
of the same function as a -m gpu test."""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # (tests/ holds the oracle-checked fuzzers: only tests may call the oracle)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

RATES = (0.5e6, 1.0e6, 1.5e6, 2.0e6, 2.7e6, 3.5e6)
NOISES = (0.05, 0.3)


def run(seeds, rates=RATES, noises=NOISES, fused=False, ctx=None, verbose=True, n_events=36000):
    import torch
    import eventcalib_amd
    from eventcalib_amd.pipeline import DetectPipeline
    import synth_stream as SS
    import test_gpu_detect as TD
    import test_gpu_events as TE
    own = ctx is None
    if own:
        ctx = eventcalib_amd.Context(0)
    pipe = DetectPipeline(ctx)
    n_ok = n_win = n_exact = n_tied = 0
    try:
        for seed, rate, noise in itertools.product(seeds, rates, noises):
            buf = SS.make_stream(n_events, rate=rate, device="cpu", seed=1000 + seed, noise_frac=noise)
            t, _, _ = SS.unpack_records(buf)
            t0, t1 = SS.tiled_windows(float(t[0]), float(t[-1]))
            pipe.set_windows(t0, t1)
            pipe.set_detect_params(5, 36, TD.THR)
            pipe.run(buf.cuda(), fused=fused)            # fused: through ecal_detect_fused_dev
            torch.cuda.synchronize()
            TE._compare(pipe, torch, buf.numpy(), t0, t1, check_labels=True)          # slicing + DBSCAN labels
            exact, tied = TD._check_windows(pipe, torch, buf.numpy(), t0, t1, 5, 36, TD.THR, exact_ties=not fused)   # extraction (the fused entry = the plain primitives)
            n_ok += 1
            n_win += len(t0)
            n_exact += exact
            n_tied += tied
            if verbose:
                print("seed %d rate %.1f noise %.2f: %d windows, max segment %d, exact %d tied %d" %
                      (seed, rate / 1e6, noise, len(t0), int(pipe.seg_cnt[:2 * len(t0)].max()), exact, tied), flush=True)
    finally:
        if own:
            ctx.close()
    return dict(streams=n_ok, windows=n_win, exact=n_exact, tied=n_tied)


if __name__ == "__main__":
    r = run(range(int(sys.argv[1]) if len(sys.argv) > 1 else 3), fused=bool(os.environ.get("FUSED")))
    print("all", r["streams"], "streams equal the oracle")
