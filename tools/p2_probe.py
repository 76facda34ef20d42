"""P2 policy only (adaptive windows, reference-like piece count) — for rocprofv3 kernel stats of one lock-step driver run."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
if os.environ.get("ECAL_AB_LIB"):        # A/B: another build of the library (ab_libs/)
    import eventcalib_amd.capi as _capi
    _capi.lib_path = lambda: os.path.abspath(os.environ["ECAL_AB_LIB"])
import eventcalib_amd, synth_stream as SS
from eventcalib_amd.adaptive import detect_keyframes, detect_keyframes_device
from eventcalib_amd.pipeline import DetectPipeline
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
pieces = int(sys.argv[2]) if len(sys.argv) > 2 else 254
nth = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ctx = eventcalib_amd.Context(0)
ev = SS.make_stream(n, device="cuda")
pipe = DetectPipeline(ctx)
dev = len(sys.argv) > 4 and sys.argv[4] in ("dev", "shared")     # policy on the device (ecal_detect_keyframes); shared = the shared-map gate
if dev:
    gm = 1 if sys.argv[4] == "shared" else 0
    ctxs = [eventcalib_amd.Context(0) for _ in range(nth)] if nth > 1 else None
    run = lambda a, b: detect_keyframes_device(ctx, ev, 5e-4, 4000, pieces, a, b, gate_mode=gm, n_threads=nth, contexts=ctxs)
else:
    run = lambda a, b: detect_keyframes(pipe, ev, 5e-4, 4000, pieces, a, b, n_threads=nth)
run(5.0, 5.5)
run(5.0, 5.0 + (n - 1) / 1e6)          # (sizes the scratch buffers: the timed run does not allocate)
torch.cuda.synchronize(); t = time.perf_counter()
kf = run(5.0, 5.0 + (n - 1) / 1e6)
torch.cuda.synchronize(); el = time.perf_counter() - t
import hashlib
print("keyframes sha", hashlib.sha1(kf["time"].tobytes() + kf["features"].tobytes()).hexdigest()[:12])
print("threads %d " % nth + "P2: %d events, %d pieces, %.4f s, %d passes (%.3f ms each), %d windows, %d keyframes" % (n, pieces, el, kf["steps"], el / kf["steps"] * 1e3, kf["windows"], len(kf["time"])))
