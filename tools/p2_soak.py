"""Soak of the keyframe search (not a test of the suite): random streams (seed, rate, noise, trajectory), random piece counts and
ranges; ecal_detect_keyframes with both gates against oracle/policy_oracle.cpp (modes 0 and 1) fed by the product's own
single-window detection.  python tools/p2_soak.py [rounds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import eventcalib_amd, eventcalib_amd.capi as capi, synth_stream as SS, oracle_lib as O
from eventcalib_amd.adaptive import detect_keyframes_device
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(2026)
ctx = eventcalib_amd.Context(0)
bad = 0
for r in range(rounds):
    seed = int(rng.integers(1, 1 << 30)); rate = float(rng.choice([1.0e6, 1.5e6, 2.0e6, 3.0e6])); noise = float(rng.choice([0.02, 0.1, 0.25]))
    SS.TRAJECTORY = str(rng.choice(["hover", "orbit"]))
    n = int(rng.choice([1_000_000, 2_000_000]))
    ev = SS.make_stream(n, rate=rate, device="cuda", seed=seed, noise_frac=noise)
    torch.cuda.synchronize()
    span = (n - 1) / rate
    t_first = 5.0 + float(rng.uniform(0, 0.3)) * span
    t_last = min(5.0 + span, t_first + float(rng.uniform(0.15, 0.45)))
    pieces = int(rng.choice([1, 2, 3, 5, 9, 17, 33, 80]))
    cache = {}
    def detect(t0, t1):
        if (t0, t1) not in cache:
            packed = capi.detect_pass(ctx, ev.data_ptr(), n, np.array([t0]), np.array([t1]), 65536, 4.0, 2, 5, 9, 4)
            found = (int(packed[0, 0]) & 0xFF) == 0 and packed[0, 1] != 0
            cache[(t0, t1)] = (found, int(packed[0, 2]), packed[0, 3:].reshape(36, 3).copy() if found else None)
        return cache[(t0, t1)]
    t0 = time.perf_counter()
    msg = []
    for mode, gm in ((0, capi.GATE_OWN_PIECE), (1, capi.GATE_SHARED_MAP)):
        ref = O.policy_run(detect, t_first, t_last, pieces, 5e-4, 4000, 9, 4, mode=mode)
        dev = detect_keyframes_device(ctx, ev, 5e-4, 4000, pieces, t_first, t_last, gate_mode=gm)
        same = dev["windows"] == ref["windows"] and all(np.array_equal(dev[k], ref[k]) for k in ("time", "duration", "events_num", "features"))
        msg.append("%s %d keyframes %d windows %s" % ("own" if mode == 0 else "shared", len(ref["time"]), ref["windows"], "ok" if same else "MISMATCH"))
        bad += 0 if same else 1
    print("round %2d seed %10d rate %.1e noise %.2f %s n %d range %.3f s pieces %2d: %s (%.1f s)" % (r, seed, rate, noise, SS.TRAJECTORY, n, t_last - t_first, pieces, "; ".join(msg), time.perf_counter() - t0), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
