#!/usr/bin/env python3
"""bench.py — M1 of BASELINE.json: Mevents/s through decode -> time-slice -> DBSCAN(+/-) -> circle
candidates on the 50 M-event synthetic circle-grid stream (configs[2]'s stream; tiled 1.5 ms
windows, policy P1 of SURVEY §8d), one process per GPU.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by the driver with torch.distributed.run (one rank per GPU, RCCL).  The metric is
quoted "on 50M-event stream @1/2/4/8 GPU": the headline is STRONG scaling — the ONE 50 M-event stream
cut into N contiguous time ranges of whole windows, one per rank, no data-path collective (SURVEY
§8e; the reference cuts the same stream into pieces, eventCameraCalib.cpp:172-179); the only
collectives are the timing barrier and a MAX over ranks.  Weak scaling (one 50 M stream per GPU) is
reported beside it.  A "step" = one full pass of the hot path over the stream, HBM-resident when the
timed region starts (the upload is never part of `value`).  Prints ONE JSON line; a failure in a leg
beside the headline is listed in `failed_legs` and makes the exit code 1.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ALGO_BYTES_PER_EVENT = 29.0     # 25 B record read once + 4 B int32 label written once
TRAFFIC_PROFILE = "r06_traffic.json"   # tools/profile_round.sh: PMC passes of this same command
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def usable_cpus():
    """Host cores this process can actually run on at once: min(hardware threads, affinity mask, cgroup CPU quota).  The GPU
    box shows 256 hardware threads behind a quota of 16 CPUs: a CPU baseline on "254 threads" there is 16 cores' worth."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


class leg:
    """`with leg(out, key):` around a leg beside the headline that only rank 0 of a one-rank run executes: an exception there is
    printed, recorded under `key` (unless the leg had already put its result there) and the JSON line still comes out."""

    def __init__(self, out, key):
        self.out, self.key = out, key

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if et is None or not issubclass(et, Exception):
            return False
        import traceback
        traceback.print_exception(et, ev, tb, file=sys.stderr)
        self.out.setdefault(self.key, {"error": "%s: %s" % (et.__name__, str(ev)[:300])})
        return True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--events", type=int, default=50_000_000)
    ap.add_argument("--cpu-sample", type=int, default=10_000_000, help="events timed on the CPU oracle (0 = skip)")
    ap.add_argument("--solver-iters", type=int, default=8, help="LM iterations timed for M2 (0 = skip the solver leg)")
    ap.add_argument("--solver-cpu-sample", type=int, default=300_000, help="residuals timed on the CPU oracle")
    ap.add_argument("--p2-pieces", type=int, default=-1,
                    help="pieces of the adaptive-window policy P2 (SURVEY 8d) measured after M1 (0 = skip; -1 = the reference's "
                         "own choice on this host, 5 * (hardware threads - 2), and 4096)")
    ap.add_argument("--no-h2d", action="store_true", help="skip the PCIe upload timing")
    ap.add_argument("--profile", action="store_true",
                    help="roctx ranges around the library's stage entry points (ecal_set_profile_ranges): run under `rocprofv3 "
                         "--marker-trace --kernel-trace --stats -- python3 bench.py --profile ...`")
    ap.add_argument("--no-fixed-cost", action="store_true",
                    help="skip the pass_ms_fixed measurement (passes over S / 2 and S / 4 windows: under a profiler they would mix "
                         "smaller launches into the kernels' average durations)")
    ap.add_argument("--event-point", action="store_true",
                    help="the timed passes also write the event -> point map (4 B per event; an output of this library's own that "
                         "only the association stage reads — the reference's EventFrame has no such member)")
    ap.add_argument("--scaling", choices=["strong", "both"], default="both",
                    help="`value` is always the strong-scaling figure (ONE --events stream cut into N time ranges: BASELINE.json's "
                         "metric is quoted on one 50 M-event stream at 1/2/4/8 GPUs); \"both\" adds, under --gpus N > 1, the "
                         "weak-scaling leg beside it (one --events stream per GPU)")
    ap.add_argument("--ingest-events", type=int, default=200_000_000,
                    help="events of the double-buffered ingest leg (configs[4]; host-resident stream; 0 = skip)")
    ap.add_argument("--calib-views", type=int, default=64,
                    help="views of the init calibration leg (configs[3]: 64 views sharded over the GPUs; 0 = skip)")
    ap.add_argument("--calib-cpu-views", type=int, default=32,
                    help="views timed on the numpy oracle (0 = skip); its dense (12 + 6V)^2 solve makes the time grow with V^2 - V^3: 32 views "
                         "are ~10 - 15 s, all 64 would be ~45 s")
    ap.add_argument("--e2e-events", type=int, default=50_000_000,
                    help="events of the end-to-end leg: one stream with tilted views through keyframe search -> init calibration -> "
                         "rectify -> splines -> ecal_associate_dev -> the spline solve fed by THAT association (0 = skip)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start one rank per GPU as CHILD processes of this one (which has
        # not touched the GPU and never will — a process that has initialised HIP must not exec), relay rank 0's JSON line
        # and exit with the launcher's return code.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    # Thread pools sized for the cores this process may actually use: the GPU box shows 256 hardware threads behind a cgroup
    # quota of 16 CPUs, and a BLAS / OpenMP pool of 256 busy-waiting threads (numpy's polyfit, torch's CPU copies) gets the whole
    # cgroup throttled for the next legs — the keyframe search, whose host side polls a counter per pass, then runs at a third
    # of its speed (measured: 0.069 s against 0.235 s for the same search in the same process).
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ.setdefault(var, str(max(1, min(usable_cpus(), 16))))
    import numpy as np
    import torch
    import torch.distributed as dist
    try:
        torch.set_num_threads(max(1, min(usable_cpus(), 16)))
    except RuntimeError:
        pass

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.e2e_events > 0 and not os.environ.get("ECAL_BENCH_NO_CPP_CHAIN"):
        # the C++ driver as a cold process on a GPU nobody else holds: before this process creates its HIP context
        global _CPP_CHAIN_IDLE
        try:
            _CPP_CHAIN_IDLE = cpp_chain_on_an_idle_gpu(args.e2e_events, 1.0e6, 5.0, 5 * max(1, (os.cpu_count() or 3) - 2))
        except Exception as e:      # (reported in the line; the legs below still run)
            _CPP_CHAIN_IDLE = {"error": repr(e)[:300]}
    # test hooks (never set by the driver): run the N > 1 code paths on a one-GPU box
    backend = os.environ.get("ECAL_BENCH_BACKEND", "nccl")          # "gloo" stages collectives through the host
    if os.environ.get("ECAL_BENCH_SINGLE_DEVICE"):
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
        local_rank = 0
    dev = torch.device("cuda", local_rank)

    import eventcalib_amd
    from eventcalib_amd.pipeline import DetectPipeline
    import synth_stream as SS

    ctx = eventcalib_amd.Context(local_rank)
    if world > 1 and backend == "nccl":
        # the library's own RCCL communicator (ecal_comm_init): rank 0's id goes to the others through the process group
        idt = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            idt = torch.frombuffer(bytearray(ctx.comm_unique_id()), dtype=torch.uint8).to(dev)
        dist.broadcast(idt, 0)
        ctx.comm_init(bytes(idt.cpu().numpy().tobytes()), rank, world)
    pipe = DetectPipeline(ctx, dev)
    if args.profile:
        ctx.set_profile_ranges(True)

    # ---- synthetic input, resident in HBM before the timed region: ONE stream of --events events (seed 12345, from t = 5 s);
    # N ranks: cut into N contiguous ranges of whole windows, one per rank (tiled windows do not overlap; the adaptive policy's
    # ranges would overlap by the longest window, 9 steps).  Every rank generates ITS range of that same stream (chunks seeded
    # by index: the same records whoever generates them).  Rate = the whole stream over the slowest rank.
    n_events = args.events
    rate = 1.0e6
    t_start = 5.0
    g0, g1 = SS.tiled_windows(t_start, t_start + (n_events - 1) / rate, 1.5e-3)
    Sg = len(g0)
    w_lo, w_hi = (Sg * rank) // world, (Sg * (rank + 1)) // world
    if world > 1:
        k_lo = max(0, int((g0[w_lo] - t_start) * rate) - 2)
        k_hi = min(n_events, int((g1[w_hi - 1] - t_start) * rate) + 3)
    else:
        k_lo, k_hi = 0, n_events
    n_local = k_hi - k_lo                                   # records in this rank's buffer (its windows + a record or two each side)
    events = SS.make_stream(n_local, rate=rate, t_start=t_start, seed=12345, device=dev, k_offset=k_lo, total=n_events)
    t0, t1 = g0[w_lo:w_hi], g1[w_lo:w_hi]
    t_first = t_start
    t_last = t_start + (n_events - 1) / rate
    pipe.set_windows(t0, t1)
    S = len(t0)

    eps, minpts = 4.0, 2                                   # example.yaml:68-71
    # an untimed pass for the workload's description (config.*); the TIMED passes get no size hints from it: max_win_events
    # = max_seg_points = 0 ("unknown"), every size tier is launched — what a first pass over new data costs
    pipe.run(events, eps, minpts)
    torch.cuda.synchronize(dev)
    assert not pipe.overflowed()
    max_win = int((pipe.win_hi[:S] - pipe.win_lo[:S]).max().item())
    max_seg = int(pipe.seg_cnt[:2 * S].max().item())
    n_points = int(pipe.seg_cnt[:2 * S].sum().item())
    n_ok = int((pipe.win_info[:S, 3] == 0).sum().item())
    n_cand = int(pipe.win_info[:S, 0].sum().item())
    n_covered = int(pipe.win_hi[S - 1].item()) - int(pipe.win_lo[0].item())     # events inside this rank's windows

    def step():
        pipe.run(events, eps, minpts)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    # per-stage HIP events on the stream the kernels are launched on (torch's current stream)
    st = torch.cuda.current_stream(dev)
    c = ctx
    pipe.set_detect_params(5, 36, ctx.circle_radius_threshold(346, 260, 9, 4, True, 5.5, 1.75))

    def staged_pass(n, Sw, marks=None, event_point=None):
        """One pass of the hot path, stage by stage, on the launch stream; marks: five HIP events recorded around the stages."""
        want_ep = args.event_point if event_point is None else event_point
        pk = pipe._pk()
        if marks:
            marks[0].record(st)
        c.window_bounds_dev(events.data_ptr(), n, pipe.t0.data_ptr(), pipe.t1.data_ptr(), Sw, pipe.win_lo.data_ptr(),
                            pipe.win_hi.data_ptr(), pipe.win_base.data_ptr(), st.cuda_stream)
        if marks:
            marks[1].record(st)
        # the three stages on packed points (ecal_packed_points): integer-pixel windows travel as 4-byte words between the
        # stages; the doubles of positiveEvents_ / negativeEvents_ are written on request only (pipe.xy), outside this loop
        # (d_event_point = NULL unless --event-point: the event -> point map is this library's own extra for the association
        # stage, the reference's EventFrame keeps none; the end-to-end leg, which associates, asks for it)
        c.slice_events_packed_dev(events.data_ptr(), n, pipe.win_lo.data_ptr(), pipe.win_hi.data_ptr(), pipe.win_base.data_ptr(), Sw, 0, n,
                                  pipe._xy.data_ptr(), pipe.seg_off.data_ptr(), pipe.seg_cnt.data_ptr(),
                                  pipe.event_point.data_ptr() if want_ep else 0, pipe.flags.data_ptr(), pk, st.cuda_stream)
        if marks:
            marks[2].record(st)
        c.dbscan_batch_packed_dev(pipe._xy.data_ptr(), pipe.seg_off.data_ptr(), pipe.seg_cnt.data_ptr(), 2 * Sw, n, 0,
                                  eps, minpts, pipe.labels.data_ptr(), pipe.n_clusters.data_ptr(), pk, st.cuda_stream)
        if marks:
            marks[3].record(st)
        # the exact extraction (the library's default): plain pass + the reference's member order for the clusters whose median
        # is tied in norm + re-extraction of their windows
        c.extract_batch_packed_dev(pipe._xy.data_ptr(), pipe.seg_off.data_ptr(), pipe.seg_cnt.data_ptr(), pipe.labels.data_ptr(),
                                   pipe.n_clusters.data_ptr(), Sw, n, eps, pipe.det[0], pipe.det[1], pipe.det[2],
                                   pipe.win_info.data_ptr(), pipe.cand_pair.data_ptr(), pipe.cand_xyr.data_ptr(),
                                   pipe.kept_labels.data_ptr(), pipe.rep.data_ptr(), pk, st.cuda_stream, fit_circle=pipe.det[3],
                                   knn_num=pipe.det[4])
        if marks:
            marks[4].record(st)

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(args.steps)]
    barrier()
    t_begin = time.perf_counter()
    for k in range(args.steps):
        staged_pass(n_local, S, ev[k])
    barrier()
    elapsed = time.perf_counter() - t_begin
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        cov = torch.tensor([float(n_covered)], dtype=torch.float64, device=dev)
        dist.all_reduce(cov, op=dist.ReduceOp.SUM)
        covered_total = int(cov.item())
    else:
        covered_total = n_covered

    ms_per_step_local = elapsed / max(args.steps, 1) * 1e3
    stage_ms = np.zeros(4)
    for k in range(args.steps):
        for j in range(4):
            stage_ms[j] += ev[k][j].elapsed_time(ev[k][j + 1])
    stage_ms /= max(args.steps, 1)

    # the part of a pass that does not shrink with the stream (launches of tiers that find their to-do lists empty, window
    # bounds, per-kernel ramp-up): passes over the first S, S/2 and S/4 windows, straight line through the three times,
    # its value at zero windows.  At N = 8 a rank's pass is an eighth of the stream: this is what caps strong scaling.
    pass_ms_fixed = None
    if rank == 0 and args.steps > 0 and S >= 64 and not args.no_fixed_cost:
        sizes, times = [S, S // 2, S // 4], []
        reps = max(10, args.steps)
        for Sw in sizes:
            for _ in range(3):
                staged_pass(n_local, Sw)
            fe = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            fe[0].record(st)
            for _ in range(reps):
                staged_pass(n_local, Sw)
            fe[1].record(st)
            torch.cuda.synchronize(dev)
            times.append(fe[0].elapsed_time(fe[1]) / reps)
        slope, icpt = np.polyfit(np.array(sizes, float), np.array(times), 1)
        pass_ms_fixed = {"ms": round(float(icpt), 4), "ms_per_1000_windows": round(float(slope) * 1e3, 5),
                         "passes_ms": {str(a): round(b, 4) for a, b in zip(sizes, times)},
                         "note": "intercept of pass time against window count (S, S/2, S/4 windows of the same stream)"}
        staged_pass(n_local, S)        # leave the arrays of the whole stream behind for the legs below
        torch.cuda.synchronize(dev)

    # the timed pass with the OTHER setting of the event -> point map (the library's DetectPipeline writes it by default, the
    # timed pass does not unless --event-point): both figures side by side, so that `value` can be compared like for like
    event_point_leg = None
    if rank == 0 and args.steps > 0:
        other = not args.event_point
        for _ in range(2):
            staged_pass(n_local, S, event_point=other)
        fe = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        fe[0].record(st)
        for _ in range(args.steps):
            staged_pass(n_local, S, event_point=other)
        fe[1].record(st)
        torch.cuda.synchronize(dev)
        other_ms = fe[0].elapsed_time(fe[1]) / args.steps
        event_point_leg = {"timed_pass_writes_the_map": bool(args.event_point),
                           "ms_per_step_with_the_map": round(ms_per_step_local if args.event_point else other_ms, 4),
                           "ms_per_step_without_the_map": round(other_ms if args.event_point else ms_per_step_local, 4),
                           "note": "d_event_point (event -> point index, 4 B per event) is this library's own output for the association "
                                   "stage; the reference's EventFrame keeps none.  DetectPipeline's default writes it; `value` is the "
                                   "setting named in config.event_point"}
        staged_pass(n_local, S)
        torch.cuda.synchronize(dev)

    plain_extract_ms = None
    if rank == 0 and args.steps > 0:
        fe = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        # the plain extraction (smaller pid at tied medians: ecal_extract_batch_dev alone), for comparison
        def plain_extract():
            c.set_median_ties(1)       # ECAL_TIES_SMALLER_PID: the plain extraction, same packed points
            try:
                c.extract_batch_packed_dev(pipe._xy.data_ptr(), pipe.seg_off.data_ptr(), pipe.seg_cnt.data_ptr(), pipe.labels.data_ptr(),
                                           pipe.n_clusters.data_ptr(), S, n_local, eps, pipe.det[0], pipe.det[1], pipe.det[2],
                                           pipe.win_info.data_ptr(), pipe.cand_pair.data_ptr(), pipe.cand_xyr.data_ptr(),
                                           pipe.kept_labels.data_ptr(), pipe.rep.data_ptr(), pipe._pk(), st.cuda_stream, fit_circle=pipe.det[3],
                                           knn_num=pipe.det[4])
            finally:
                c.set_median_ties(0)
        plain_extract()
        fe[0].record(st)
        for _ in range(args.steps):
            plain_extract()
        fe[1].record(st)
        torch.cuda.synchronize(dev)
        plain_extract_ms = fe[0].elapsed_time(fe[1]) / args.steps

    total_events = n_events * args.steps                   # every event of the ONE stream once per step, whatever N
    value = total_events / elapsed / 1e6
    ms_per_step = elapsed / max(args.steps, 1) * 1e3

    # ---- beside the headline under N > 1: the adaptive-window search of the same stream cut over the ranks, and WEAK scaling
    sharded_p2 = weak = None
    if world > 1 and args.steps > 0:
        if args.p2_pieces != 0:
            # the reference driver's adaptive-window search (policy P2) over the same stream: rank r searches the pieces
            # [P r / N, P (r + 1) / N) — with the bounds they have in the whole run (ecal_adaptive_params.piece_first / piece_count) —
            # on its own range of the stream.  With the reference's gate (one keyframe map, single-worker order: the 1-GPU front
            # ends' default) a rank needs ONE frame from the ranks before it in time: a chain of N - 1 messages of 66 doubles
            # (ecal_detect_keyframes_sharded, DistHandover), its pieces running as a speculation until the frame arrives — the
            # ranks' keyframes together are the single-GPU search's, record for record.  The own-piece gate (no exchange at all)
            # is timed beside it.
            from eventcalib_amd.adaptive import detect_keyframes_device, DistHandover
            from eventcalib_amd import capi as _capi
            P = 5 * max(1, (os.cpu_count() or 3) - 2) if args.p2_pieces < 0 else args.p2_pieces
            P = max(P, world)
            tf, tl = t_first, t_last
            p_lo, p_hi = (P * rank) // world, (P * (rank + 1)) // world
            pstep = (tl - tf) / P
            q_lo = max(0, int((tl - pstep * p_hi - t_start) * rate) - 4)        # (piece 0 is the LAST in time)
            q_hi = min(n_events, int((tl - pstep * p_lo - t_start) * rate) + 5)
            del events
            torch.cuda.empty_cache()
            ev_p = SS.make_stream(q_hi - q_lo, rate=rate, t_start=t_start, seed=12345, device=dev, k_offset=q_lo, total=n_events)
            def sharded_search(gate, tag):
                # (rank r holds pieces [p_lo, p_hi): piece 0 is the LAST in time, so the rank before r in time is r + 1)
                def once(t):
                    ho = DistHandover(rank + 1 if rank + 1 < world else None, rank - 1 if rank > 0 else None, tag=t) if gate == _capi.GATE_SHARED_MAP else None
                    return detect_keyframes_device(ctx, ev_p, 5e-4, 4000, P, tf, tl, eps, minpts, piece_first=p_lo, piece_count=p_hi - p_lo,
                                                   gate_mode=gate, handover=ho)
                once(tag)   # warm-up
                barrier()
                tb = time.perf_counter()
                kp = once(tag + 1)
                barrier()
                el_p = time.perf_counter() - tb
                agg = torch.tensor([float(len(kp["time"])), float(kp["windows"])], dtype=torch.float64, device=dev)
                dist.all_reduce(agg, op=dist.ReduceOp.SUM)
                mx = torch.tensor([el_p], dtype=torch.float64, device=dev)
                dist.all_reduce(mx, op=dist.ReduceOp.MAX)
                every = [None] * world
                dist.all_gather_object(every, kp["time"].tobytes())
                import hashlib
                digest = hashlib.sha1(np.sort(np.frombuffer(b"".join(every), np.float64)).tobytes()).hexdigest()[:16]
                return {"value": round(n_events / float(mx[0].item()) / 1e6, 1), "unit": "Mevents/s", "scaling": "strong", "pieces": P,
                        "pieces_per_gpu": p_hi - p_lo, "seconds": round(float(mx[0].item()), 4), "keyframes": int(agg[0].item()),
                        "windows_evaluated": int(agg[1].item()), "keyframe_times_sha1_16": digest}
            sharded_p2 = sharded_search(_capi.GATE_SHARED_MAP, 100)
            sharded_p2.update({"gate": "shared map, single worker (the single-GPU front ends' default; the reference's TrackingBase / "
                                       "EventCalibIni::track semantics)",
                               "exchange": "%d messages of 66 doubles per search: the frame behind a rank's pieces (last keyframe's time stamp + "
                                           "row directions) to the next rank in time; pieces run speculatively until it arrives" % (world - 1),
                               "note": "the pieces of one search cut over the ranks, every rank on its own time range of the stream; the "
                                       "ranks' keyframes together == the 1-GPU line's policy_p2[gate = shared map] at the same piece count "
                                       "(keyframe_times_sha1_16 there)"})
            sharded_p2["own_piece_gate"] = sharded_search(_capi.GATE_OWN_PIECE, 200)
            sharded_p2["own_piece_gate"]["note"] = ("every piece's first success ungated: no exchange at all, not what a run of the "
                                                    "reference computes; compare with policy_p2[gate = own piece] of the 1-GPU line")
            events = ev_p
        if args.scaling == "both":
            # weak scaling: every rank its OWN --events stream (its own time range of the motion), per-GPU work fixed
            del events
            torch.cuda.empty_cache()
            tw = t_start + rank * (n_events / rate + 1.0)
            events = SS.make_stream(n_events, rate=rate, t_start=tw, seed=12345 + rank, device=dev)
            a0, a1 = SS.tiled_windows(tw, tw + (n_events - 1) / rate, 1.5e-3)
            pipe.set_windows(a0, a1)
            pipe.run(events, eps, minpts)          # (sizes the pipeline's arrays for a whole stream per rank)
            for _ in range(max(1, args.warmup)):
                staged_pass(n_events, len(a0))
            barrier()
            tb = time.perf_counter()
            for _ in range(args.steps):
                staged_pass(n_events, len(a0))
            barrier()
            el_w = time.perf_counter() - tb
            mx = torch.tensor([el_w], dtype=torch.float64, device=dev)
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            el_w = float(mx[0].item())
            weak = {"value": round(n_events * world * args.steps / el_w / 1e6, 3), "unit": "Mevents/s", "scaling": "weak",
                    "events_per_gpu": n_events, "windows_per_gpu": len(a0), "ms_per_step": round(el_w / args.steps * 1e3, 4),
                    "note": "one %dM-event stream PER GPU (per-GPU work fixed); reported beside the headline, never `value`" % (n_events // 1_000_000)}

    out = {
        "metric": "Mevents/s DBSCAN+detect",
        "value": round(value, 3),
        "unit": "Mevents/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "ONE %dM-event synthetic circle-grid stream (BASELINE configs[2]'s stream), 346x260, 1 Mev/s, tiled 1.5 ms "
                        "windows (%d windows)%s, eps 4 minpts 2; stages: window bounds + EventFrame slicing (reference point order) + "
                        "DBSCAN(+/-) + circle candidates (reference tie picks)%s; HBM-resident when the timed region starts, upload "
                        "excluded (reported beside: h2d, ingest)"
                        % (n_events // 1_000_000, Sg, "" if world == 1 else ", cut into %d time ranges of whole windows, one per GPU" % world,
                           " + the event -> point map" if args.event_point else ""),
            "events_total": n_events, "events_covered_by_the_ranks_windows": covered_total, "windows_total": Sg,
            "events_per_gpu": n_covered, "windows_per_gpu": S, "unique_points_per_gpu": n_points,
            "max_window_events": max_win, "max_segment_points": max_seg,
            "windows_reaching_pairing": n_ok, "circle_candidates": n_cand,
            "sharding": "time ranges of ONE stream, no data-path collective" if world > 1 else "single GPU",
            "event_point": bool(args.event_point),
        },
    }
    if event_point_leg is not None:
        out["event_point_map"] = event_point_leg
    if weak is not None:
        out["weak_scaling"] = weak
    if sharded_p2 is not None:
        out["policy_p2_sharded"] = sharded_p2
    if pass_ms_fixed is not None:
        out["pass_ms_fixed"] = pass_ms_fixed
    if rank == 0:
        # the dominant KERNEL: the extraction stage is its first pass (which resolves the ties of small clusters itself since round 5)
        # + the launches of the listed path for what that leaves; the plain pass — timed alone below the stage — stands for it
        kernel_ms = [float(x) for x in stage_ms]
        if plain_extract_ms is not None:
            kernel_ms[3] = float(plain_extract_ms)
        dom = int(np.argmax(kernel_ms))
        names = ["window_bounds_base_kernel", "slice_hash_ref_kernel", "dbscan_pixel_kernel", "extract_kernel"]   # reference point order (the default)
        achieved = ALGO_BYTES_PER_EVENT * n_covered / (kernel_ms[dom] * 1e-3) / 1e9      # rank 0's launch: the events of ITS windows
        # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process; the
        # committed profile (tools/pmc_traffic.py over two rocprofv3 --pmc passes of this same command) is
        # quoted when it was taken on the same workload size, else null
        traffic = pass_traffic = None
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", TRAFFIC_PROFILE)))
            if tr.get("events") == n_events:
                # the detection kernels of one timed pass: every kernel of the profile's passes (a pass = one launch of
                # window_bounds_kernel) except the other legs' — the fused pass, the plain extraction (MODE 0), solver, calibration
                passes = max(1, max(v.get("launches", 0) for k, v in tr["kernels"].items() if "window_bounds" in k))
                skip = ("normal_eq", "reduce_heads", "calib_", "view_", "residual_rows", "arrow_", "lm_plus", "solver_", "bucket_table",
                        "associate", "scan_blocks", "extract_kernel<false, 0>", "extract_list_kernel<false, 0>",
                        "extract_first_list_kernel<false, 0>", "grid_order", "adaptive_", "gather_features", "rectify", "pnp_", "sort_")
                pass_traffic = sum(v["hbm_bytes_per_launch"] * min(1.0, v.get("launches", passes) / passes) for k, v in tr["kernels"].items()
                                   if not any(x in k for x in skip))
            key = {2: "ecal::dbscan_pixel_kernel", 1: "ecal::slice_hash_ref_kernel", 3: "ecal::extract_kernel"}.get(dom)
            hits = [v for k, v in tr["kernels"].items() if k.split("<")[0] == key]   # template arguments vary
            if tr.get("events") == n_events and hits:
                traffic = max(h["hbm_bytes_per_launch"] for h in hits)
        except Exception:
            traffic = None
        out["roofline"] = {
            "bound": "hbm", "kernel": names[dom], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
            "algorithmic_bytes_per_launch": ALGO_BYTES_PER_EVENT * n_covered,
            "kernel_ms": round(kernel_ms[dom], 4),
            # the whole pass (every kernel of the timed region) on the same algorithmic bytes, and its measured traffic
            "whole_pass": {"achieved": round(ALGO_BYTES_PER_EVENT * n_events / (ms_per_step * 1e-3) / 1e9, 2),
                           "frac": round(ALGO_BYTES_PER_EVENT * n_events / (ms_per_step * 1e-3) / 1e9 / (HBM_PEAK_GBS * world), 5),
                           "peak": HBM_PEAK_GBS * world,
                           "traffic": pass_traffic,
                           "traffic_over_algorithmic": round(pass_traffic / (ALGO_BYTES_PER_EVENT * n_events), 3) if pass_traffic else None},
            "stage_ms": {"window_bounds": round(float(stage_ms[0]), 4), "slice": round(float(stage_ms[1]), 4),
                         "dbscan": round(float(stage_ms[2]), 4), "extract": round(float(stage_ms[3]), 4)},
        }
        if plain_extract_ms is not None:
            out["median_ties"] = {"timed": "reference (std::nth_element over the reference's cluster member order where a median is tied)",
                                  "extract_ms_exact": round(float(stage_ms[3]), 4), "extract_ms_smaller_pid_rule": round(float(plain_extract_ms), 4),
                                  "pass_ms_smaller_pid_rule": round(float(stage_ms[0] + stage_ms[1] + stage_ms[2] + plain_extract_ms), 4)}
        if args.cpu_sample > 0 and world == 1:
            with leg(out, "cpu_baseline"):
                import oracle_lib as O
                m = min(args.cpu_sample, n_events)
                rec = events[: m * 25].cpu().numpy()
                nw = int(np.searchsorted(t1, t_start + (m - 1) / rate))
                tc = time.perf_counter()
                cev, ccl = O.detect_windows(rec, t0[:nw], t1[:nw], eps, minpts)
                cel = time.perf_counter() - tc
                # the reference's own threading: T = hardware threads - 2 workers over 5 T pieces (eventCameraCalib.cpp:172-190).
                # The headline runs min(T, the CPUs this process may use) threads — the fair run on a box whose cgroup quota is below
                # its hardware thread count; the reference's literal T (oversubscribed under such a quota) is reported beside it.
                # Bounded samples: as many windows as the threads finish in about the single-thread sample's time.
                T = max(1, (os.cpu_count() or 3) - 2)
                quota = usable_cpus()

                def threaded(n_thr):
                    nw_mt = int(min(len(t0), nw * min(n_thr, 16)))
                    rec_mt = events[: min(n_events, int((t1[nw_mt - 1] - t_start) * rate) + 2) * 25].cpu().numpy()
                    tc = time.perf_counter()
                    mev, _ = O.detect_windows_mt(rec_mt, t0[:nw_mt], t1[:nw_mt], eps, minpts, n_thr)
                    mel = time.perf_counter() - tc
                    return nw_mt, mev, mel
                Tq = max(1, min(T, quota))
                nw_mt, mev, mel = threaded(Tq)
                literal = None
                if T != Tq:
                    nw_l, mev_l, mel_l = threaded(T)
                    literal = {"value": round(mev_l / mel_l / 1e6, 4), "unit": "Mevents/s", "threads": T, "cores": Tq,
                               "sample": "first %d windows (%d events), %.2f s" % (nw_l, mev_l, mel_l),
                               "note": "the reference's literal hardware_concurrency() - 2 threads inside this process's CPU quota"}
                out["cpu_baseline"] = {
                    "value": round(mev / mel / 1e6, 4), "unit": "Mevents/s", "cores": Tq, "threads": Tq,
                    "host_cpu_quota": quota, "kind": "port",
                    "sample": "first %d windows (%d events) of the same stream, oracle EventFrame + extractFeatures (DBSCAN +/-, "
                              "filter, medians, pairing) per window on %d threads over %d pieces (the reference driver's "
                              "threading on the CPUs this process may use), %.2f s" % (nw_mt, mev, Tq, 5 * Tq, mel),
                    "single_thread": {"value": round(cev / cel / 1e6, 4), "unit": "Mevents/s", "cores": 1,
                                      "sample": "first %d windows (%d events), %.1f s" % (nw, cev, cel)},
                    "reference_thread_count": literal,
                    "host_cpus": os.cpu_count(),
                }
    # ---- reported beside the contract number (rank 0, one GPU): PCIe upload and the reference's window policy ----
    if rank == 0 and world == 1 and not args.no_h2d:
        with leg(out, "h2d"):
            host = torch.empty(events.numel(), dtype=torch.uint8, pin_memory=True)
            host.copy_(events)
            dst = torch.empty_like(events)
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            dst.copy_(host, non_blocking=True)
            e1.record(st)
            torch.cuda.synchronize(dev)
            h2d_ms = e0.elapsed_time(e1)
            del host, dst
            # SURVEY 8(d)'s counting rule ("H2D copy included and also reported separately"): the rate with the one upload of the
            # stream charged to the pass — beside `value`, which is HBM-resident by bench.py's contract
            out["value_including_upload"] = round(n_events / ((h2d_ms + ms_per_step) * 1e-3) / 1e6, 1)
            out["h2d"] = {"bytes": int(events.numel()), "ms": round(h2d_ms, 3),
                          "GBs": round(events.numel() / h2d_ms / 1e6, 2),
                          "Mevents_per_s_including_upload": round(n_events / ((h2d_ms + ms_per_step) * 1e-3) / 1e6, 1),
                          "note": "pinned host -> HBM copy of the packed stream, once per stream; never part of `value`"}
    if rank == 0 and world == 1 and args.p2_pieces != 0:
        with leg(out, "policy_p2_error"):
            # policy P2 of SURVEY 8d: the reference driver's adaptive success / slide / grow windows per piece
            # (eventCameraCalib.cpp:49-95) + keyframe gate, lock-step over all pieces; counted as the reference
            # would: all events of the stream / wall time, although the policy skips frameGap after every keyframe.
            # The number of pieces is a property of the HOST in the reference (5 * (hardware threads - 2)); the chain of
            # dependent windows inside a piece is what serialises, so the GPU wants many short pieces.
            from eventcalib_amd.adaptive import detect_keyframes, detect_keyframes_device
            ref_pieces = 5 * max(1, (os.cpu_count() or 3) - 2)
            runs = [ref_pieces, 4096] if args.p2_pieces < 0 else [args.p2_pieces]
            out["policy_p2"] = []
            for pieces in runs:
                nth = 2   # more host threads do not help: a pass is bound by the GPU-side chain of its largest window
                # (warm-ups over the WHOLE stream: a context sizes its scratch buffers by the call, and on a box whose memory another
                # process has just given back a first full-size call has been seen to spend 0.15 - 0.9 s in its allocations)
                detect_keyframes(pipe, events, 5e-4, 4000, pieces, t_first, t_last, eps, minpts, n_threads=nth)   # warm-up
                torch.cuda.synchronize(dev)
                tp = time.perf_counter()
                kf = detect_keyframes(pipe, events, 5e-4, 4000, pieces, t_first, t_last, eps, minpts, n_threads=nth)
                torch.cuda.synchronize(dev)
                p2_s = time.perf_counter() - tp
                out["policy_p2"].append({"value": round(n_events / p2_s / 1e6, 1), "unit": "Mevents/s", "seconds": round(p2_s, 4),
                                         "pieces": pieces, "host_threads": nth, "lockstep_passes": kf["steps"], "windows_evaluated": kf["windows"],
                                         "keyframes": int(len(kf["time"])),
                                         "driver": "host",
                                         "note": "adaptive windows + grid ordering + keyframe gate, host-driven (one H2D of the window "
                                                 "bounds and one D2H of verdicts + ordered circles per pass)"})
                # the same policy with the rule on the device (ecal_detect_keyframes): no per-pass host round trip
                from eventcalib_amd import capi as _capi
                detect_keyframes_device(ctx, events, 5e-4, 4000, pieces, t_first, t_last, eps, minpts,
                                        gate_mode=_capi.GATE_OWN_PIECE)   # warm-up
                torch.cuda.synchronize(dev)
                tp = time.perf_counter()
                kd = detect_keyframes_device(ctx, events, 5e-4, 4000, pieces, t_first, t_last, eps, minpts, gate_mode=_capi.GATE_OWN_PIECE)
                p2d_s = time.perf_counter() - tp
                out["policy_p2"].append({"value": round(n_events / p2d_s / 1e6, 1), "unit": "Mevents/s", "seconds": round(p2d_s, 4),
                                         "pieces": pieces, "host_threads": 1, "longest_window_chain": kd["steps"],
                                         "windows_evaluated": kd["windows"],
                                         "keyframes": int(len(kd["time"])), "driver": "device",
                                         "same_keyframes_as_host_driver": bool(np.array_equal(kd["time"], kf["time"])),
                                         "note": "the same policy in one call: five stages + grid ordering over every piece's current window and "
                                                 "the windows that follow it if every verdict is the likely one (the slots of a pass go to the pieces "
                                                 "still at work) + one policy kernel that applies the rule along that chain while the verdicts agree, "
                                                 "enqueued back to back; the host follows a 4-byte counter two passes behind",
                                         "gate": "own piece"})
                # ... and with the reference's own gate semantics: ONE keyframe map, single-worker order (ECAL_GATE_SHARED_MAP ==
                # oracle/policy_oracle.cpp mode 1): speculation as above + verification rounds across the piece boundaries
                from eventcalib_amd import capi as _capi
                detect_keyframes_device(ctx, events, 5e-4, 4000, pieces, t_first, t_last, eps, minpts,
                                        gate_mode=_capi.GATE_SHARED_MAP)   # warm-up, as the two timings above (first-call allocations)
                torch.cuda.synchronize(dev)
                tp = time.perf_counter()
                ks = detect_keyframes_device(ctx, events, 5e-4, 4000, pieces, t_first, t_last, eps, minpts, gate_mode=_capi.GATE_SHARED_MAP)
                p2s_s = time.perf_counter() - tp
                out["policy_p2"].append({"value": round(n_events / p2s_s / 1e6, 1), "unit": "Mevents/s", "seconds": round(p2s_s, 4),
                                         "pieces": pieces, "host_threads": 1, "longest_window_chain": ks["steps"],
                                         "windows_evaluated": ks["windows"], "keyframes": int(len(ks["time"])), "driver": "device",
                                         "keyframe_times_sha1_16": __import__("hashlib").sha1(np.sort(ks["time"]).tobytes()).hexdigest()[:16],
                                         "gate": "shared map, single worker (the reference's TrackingBase / EventCalibIni::track semantics)"})
            pipe.set_windows(t0, t1)
    # ------------------------------------------------------------------------------------------
    # M2: Levenberg-Marquardt iterations/s of the continuous-time solve on the same stream
    # (configs[2]: 50 M events -> ~45 M associated residuals, control point every 50 steps = 25 ms)
    # ------------------------------------------------------------------------------------------
    def guarded(fn, *a):
        """The legs beside the headline: with one rank a failure there is reported in its place and the line still comes out
        (with several ranks a leg's collectives have to be entered by all of them: an error stays an error)."""
        if world > 1:
            return fn(*a)
        try:
            return fn(*a)
        except Exception as exc:   # noqa: BLE001
            import traceback
            traceback.print_exc(file=sys.stderr)
            return {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}

    if args.solver_iters > 0:
        del events
        pipe = None
        torch.cuda.empty_cache()
        out_solver = guarded(solver_leg, args, ctx, dev, world, rank, n_events, rate, t_start, dist, torch, np)
        if rank == 0:
            out["solver"] = out_solver
    if args.ingest_events > 0:
        out_ingest = guarded(ingest_leg, args, ctx, dev, world, rank, dist, torch, np, rate)
        if rank == 0:
            out["ingest"] = out_ingest
    if args.calib_views > 0:
        out_calib = guarded(calib_leg, args, ctx, dev, world, rank, dist, torch, np)
        if rank == 0:
            out["init_calibration"] = out_calib
    if args.e2e_events > 0 and rank == 0 and world == 1:
        out["end_to_end"] = guarded(e2e_leg, args, ctx, dev, torch, np)
    failed = []
    if rank == 0:
        # a leg beside the headline that raised (a correctness assertion inside it, a library error) left {"error": ...} in its
        # place: the line still comes out, the legs are named at the top level and the process exits 1
        def walk(prefix, node):
            if isinstance(node, dict):
                if "error" in node and len(node) == 1:
                    failed.append(prefix)
                for k, v in node.items():
                    walk(prefix + "." + k if prefix else k, v)
            elif isinstance(node, list):
                for i, v in enumerate(node):
                    walk("%s[%d]" % (prefix, i), v)
        walk("", out)
        out["failed_legs"] = failed
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    ctx.close()
    if failed:
        sys.exit(1)


def e2e_leg(args, ctx, dev, torch, np):
    """The reference driver's chain on one stream (eventCameraCalib.cpp:99-233): the solver's residuals here are what
    ecal_associate_dev finds in the event stream (EventCalibSpline.cpp:140-192), not generated records."""
    import synth_stream as SS
    from eventcalib_amd.calibrate import calibrate_stream
    n, rate, t_start = args.e2e_events, 1.0e6, 5.0
    SS.TRAJECTORY = "orbit"        # tilted views: a near fronto-parallel sequence leaves the focal length unobservable
    try:
        ev = SS.make_stream(n, rate=rate, t_start=t_start, device=dev, seed=21)
    finally:
        SS.TRAJECTORY = "hover"
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    # the reference's piece count on this host: 5 * (hardware threads - 2) (eventCameraCalib.cpp:172-173)
    pieces = 5 * max(1, (os.cpu_count() or 3) - 2)
    from eventcalib_amd import capi as _capi
    calibrate_stream(ctx, ev, t_start, t_start + (n - 1) / rate, piece_num=pieces)   # warm-up (as the timed loop's): scratch of the context at its final sizes; the C++ chain below is the cold process
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    # the front ends' default: the reference's single-worker keyframe gate (one keyframe map for all pieces, design/11_keyframe_gate.md)
    r = calibrate_stream(ctx, ev, t_start, t_start + (n - 1) / rate, piece_num=pieces)
    wall = time.perf_counter() - t0
    sp = r["spline"]
    # the same chain with the schedule-free own-piece gate (every piece's first success ungated): the fast option
    calibrate_stream(ctx, ev, t_start, t_start + (n - 1) / rate, piece_num=pieces, gate_mode=_capi.GATE_OWN_PIECE)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    rs = calibrate_stream(ctx, ev, t_start, t_start + (n - 1) / rate, piece_num=pieces, gate_mode=_capi.GATE_OWN_PIECE)
    wall_s = time.perf_counter() - t0
    own = {"keyframes": rs["keyframes"], "wall_seconds_whole_chain": round(wall_s, 3),
           "keyframe_search_seconds": round(rs["stage_seconds"]["keyframe_search"], 4),
           "refined_fx_rel_err": float(abs(rs["intrinsics"][0] / SS.FX - 1)), "lm_iterations": rs["spline"]["iterations"],
           "gate": "own piece (every piece's first success ungated: schedule-free, not what a run of the reference computes)"}
    cpp_beside = cpp_chain(ev, n, rate, t_start, pieces, np) if not os.environ.get("ECAL_BENCH_NO_CPP_CHAIN") else None
    cpp = _CPP_CHAIN_IDLE if _CPP_CHAIN_IDLE is not None else cpp_beside
    # configs[4]'s camera through the same chain: Kannala-Brandt stream, fisheye init calibration / PnP / rectify / spline residual
    del ev
    SS.TRAJECTORY, SS.CAMERA = "orbit", "fisheye"
    try:
        evf = SS.make_stream(n, rate=rate, t_start=t_start, device=dev, seed=21)
    finally:
        SS.TRAJECTORY, SS.CAMERA = "hover", "pinhole"
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    rf = calibrate_stream(ctx, evf, t_start, t_start + (n - 1) / rate, piece_num=pieces, fisheye=True)
    wall_f = time.perf_counter() - t0
    del evf
    bf = rf["intrinsics"][4:9]
    th = np.linspace(0.02, 0.30, 40)
    thd = th * (1 + SS.KB[0] * th ** 2 + SS.KB[1] * th ** 4 + SS.KB[2] * th ** 6 + SS.KB[3] * th ** 8)
    back = thd * (1 + bf[0] * thd ** 2 + bf[1] * thd ** 4 + bf[2] * thd ** 6 + bf[3] * thd ** 8 + bf[4] * thd ** 10)
    fish = {"events": n, "keyframes": rf["keyframes"], "residuals_from_association": rf["spline"]["residuals"],
            "lm_iterations": rf["spline"]["iterations"], "init_fx_rel_err": float(abs(rf["init"]["intr"][0] / SS.FX - 1)),
            "refined_fx_rel_err": float(abs(rf["intrinsics"][0] / SS.FX - 1)),
            "refined_cx_err_px": float(abs(rf["intrinsics"][2] - (SS.CX - 0.5))),
            "angle_map_max_err_rad_within_0.3rad": float(np.abs(back - th).max()), "wall_seconds_whole_chain": round(wall_f, 3),
            "stage_seconds": {k: round(v, 4) for k, v in rf["stage_seconds"].items()},
            "note": "Kannala-Brandt stream (k = 0.05, -0.01, 0.002, 0); fisheye model in the init calibration (started from the radial "
                    "model's focal length), PnP, rectifyFeatures' projections and the spline residual (new functionality)"}
    return {"cpp_chain": cpp, "cpp_chain_beside_this_process": cpp_beside, "fisheye": fish, "own_piece_gate": own, "gate": "shared map, single worker (TrackingBase.cpp:16-46, EventCalibIni.cpp:26-36): the front ends' default", "events": n, "keyframes": r["keyframes"], "init_fx_rel_err": float(abs(r["init"]["intr"][0] / SS.FX - 1)),
            "residuals_from_association": sp["residuals"], "unknowns": sp["unknowns"], "splines": sp["splines"],
            "lm_iterations": sp["iterations"], "lm_seconds": round(sp["seconds"], 4),
            "lm_iterations_per_s": round(sp["iterations"] / max(sp["seconds"], 1e-9), 2),
            "refined_fx_rel_err": float(abs(r["intrinsics"][0] / SS.FX - 1)),
            "refined_cx_err_px": float(abs(r["intrinsics"][2] - (SS.CX - 0.5))), "wall_seconds_whole_chain": round(wall, 3),
            "pieces": pieces, "stage_seconds": {k: round(v, 4) for k, v in r["stage_seconds"].items()},
            "note": "keyframe search (policy P2) -> init calibration -> PnP / checkPose / rectify -> spline fit -> association of "
                    "every event -> LM; host-side Python glue between the stages is inside wall_seconds_whole_chain; second run in "
                    "this process (the first sizes the context's scratch buffers) - cpp_chain is a cold process"}


CHAIN_YAML = """%%YAML:1.0
StartTime: %(start)s
MotionTimeStep: 5e-4
FrameEventNumThreshold: 4000
Camera.width: 346
Camera.height: 260
Is_Pattern_Asymmetric: 1
BoardSize_Rows: 9
BoardSize_Cols: 4
Square_Size: 5.5
Circles_Radius: 1.75
Calibrate_NrOfFrameToUse: 200
Calibrate_UseFisheyeModel: 0
Calibrate_FixAspectRatio: 1
Calibrate_AssumeZeroTangentialDistortion: 1
Calibrate_FixPrincipalPointAtTheCenter: 1
Fix_K1: 0
Fix_K2: 0
Fix_K3: 0
Fix_K4: 1
Fix_K5: 1
dbscan_eps: 4
dbscan_startMinSample: 2
clusterMinSample: 5
knn_num: 3
fitCircle: 0
useSO3: 0
reduceMap: 0
PieceNum: %(pieces)d
"""


def _run_driver(tmp, exe, note):
    """One cold run of the C++ driver on tmp/events.bin + tmp/settings.yaml: wall seconds of the process, its own stage seconds."""
    import subprocess

    def throttled():
        # the cgroup's CPU-quota statistics (v2: cpu.stat throttled_usec; v1: cpu/cpu.stat throttled_time in ns): a child that
        # shares a throttled cgroup with this process's thread pools loses wall time that is not its own
        for path, key, div in (("/sys/fs/cgroup/cpu.stat", "throttled_usec", 1e3), ("/sys/fs/cgroup/cpu/cpu.stat", "throttled_time", 1e6),
                               ("/sys/fs/cgroup/cpu,cpuacct/cpu.stat", "throttled_time", 1e6)):
            try:
                for ln in open(path):
                    if ln.startswith(key):
                        return float(ln.split()[1]) / div
            except OSError:
                pass
        return None
    thr0 = throttled()
    t0 = time.perf_counter()
    out = subprocess.run([exe, os.path.join(tmp, "settings.yaml"), os.path.join(tmp, "events.bin"), tmp, "batch"],
                         capture_output=True, text=True, timeout=900)
    wall = time.perf_counter() - t0
    thr1 = throttled()
    if out.returncode != 0:
        return {"error": (out.stdout + out.stderr)[-400:]}
    stages = {l.split()[1]: round(float(l.split()[2]), 4) for l in out.stdout.splitlines() if l.startswith("stage ")}
    lines = [l for l in out.stdout.splitlines() if not l.startswith("stage ")]
    ref = lines[2].split()
    after = sum(v for k, v in stages.items() if k not in ("runtime_init", "load_file", "upload"))
    return {"process_wall_seconds": round(wall, 3), "stage_seconds": stages, "seconds_after_upload": round(after, 4),
            "keyframes": int(lines[0].split()[1]), "refined_fx": float(ref[1]), "residuals": int(ref[11]),
            "lm_iterations": int(ref[13]), "splines": int(ref[15]),
            "cgroup_throttled_ms_during_the_run": None if thr0 is None or thr1 is None else round(thr1 - thr0, 1),
            "load_average": [round(v, 1) for v in os.getloadavg()], "note": note}


def _driver_exe():
    import subprocess
    root = os.path.dirname(os.path.abspath(__file__))
    # the product's executable, built with the library (eventcalib_amd/csrc/Makefile, target `driver`)
    exe = os.path.join(root, "eventcalib_amd", "unit_test_eventCameraCalib")
    if not os.path.exists(exe):      # (it travels to the GPU box prebuilt, like libecal.so)
        cc = subprocess.run(["make", "-s", "-C", os.path.join(root, "eventcalib_amd", "csrc"), "driver"], capture_output=True, text=True)
        if cc.returncode != 0 or not os.path.exists(exe):
            return None, "make driver: " + (cc.stdout + cc.stderr)[-300:]
    return exe, None


_CPP_CHAIN_IDLE = None


def cpp_chain_on_an_idle_gpu(n, rate, t_start, pieces):
    """The drop-in as a user runs it: eventcalib_amd/unit_test_eventCameraCalib (host/event_camera_calib_main.cpp: the reference
    driver's main on the C++ shims, argv = settings.yaml events.bin saveDir as eventCameraCalib.cpp:105-110) as a cold process on a
    GPU NO OTHER PROCESS HOLDS.  Called at the very start of this benchmark, before this process creates its own HIP context: the
    stream file is written by a helper process that exits (the same generator, the same seed as the end-to-end leg's stream), then
    the driver runs twice (the first run of a binary on a box also pays for its page faults).  Beside a process that keeps a
    context with many queues alive — this benchmark, later — the driver's launch-bound stages take twice as long (the GPU's
    scheduler shares the queues of both processes): that figure is reported too (cpp_chain_beside_this_process)."""
    import shutil
    import subprocess
    import tempfile
    exe, err = _driver_exe()
    if exe is None:
        return {"error": err}
    root = os.path.dirname(os.path.abspath(__file__))
    tmp = tempfile.mkdtemp(prefix="ecal_chain_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)   # the stream: a RAM-backed file
    try:
        helper = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import synth_stream as SS; SS.TRAJECTORY = 'orbit'; "
                  "ev = SS.make_stream(%d, rate=%r, t_start=%r, device='cuda', seed=21); ev.cpu().numpy().tofile(%r)"
                  % (root, os.path.join(root, "tests"), n, rate, t_start, os.path.join(tmp, "events.bin")))
        h = subprocess.run([sys.executable, "-c", helper], capture_output=True, text=True, timeout=900)
        if h.returncode != 0:
            return {"error": "stream helper: " + (h.stdout + h.stderr)[-300:]}
        open(os.path.join(tmp, "settings.yaml"), "w").write(CHAIN_YAML % dict(start=t_start, pieces=pieces))
        note = ("process start to exit incl. HIP runtime initialisation and EventContainer::loadFile (ecal_stream_create_from_file: "
                "1.25 GB from a RAM-backed file, chunked reads overlapped with the upload); no other process on the GPU")
        first = _run_driver(tmp, exe, note)
        second = _run_driver(tmp, exe, note)
        if "error" in first or "error" in second:
            return first if "error" in first else second
        best = dict(second)
        # (which run is which: the headline fields are the SECOND run — a new process on an idle GPU with the binary's pages and the
        # code objects warm in the page cache; the box's very first run, the coldest figure, is first_run_on_this_box; the run
        # beside this benchmark's own context — what rounds 1 - 4 called the cold process — is cpp_chain_beside_this_process)
        best["which_run"] = "second run of the binary on this box: new process, idle GPU, warm page cache"
        best["first_run_on_this_box"] = {"process_wall_seconds": first["process_wall_seconds"], "stage_seconds": first["stage_seconds"],
                                         "which_run": "first run of the binary on this box: cold page cache, cold code-object load"}
        return best
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cpp_chain(ev, n, rate, t_start, pieces, np):
    """The same executable on the same stream while THIS process is alive with its HIP context, streams and worker pools (what
    earlier rounds reported as the cold process): its own per-stage seconds.  The stream is a file here, so loading and uploading
    it are stages of their own; `seconds_after_upload` is what compares with the Python chain's wall time."""
    import shutil
    import tempfile
    exe, err = _driver_exe()
    if exe is None:
        return {"error": err}
    tmp = tempfile.mkdtemp(prefix="ecal_chain_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)   # the stream: a RAM-backed file
    try:
        ev.cpu().numpy().tofile(os.path.join(tmp, "events.bin"))
        open(os.path.join(tmp, "settings.yaml"), "w").write(CHAIN_YAML % dict(start=t_start, pieces=pieces))
        time.sleep(float(os.environ.get("ECAL_BENCH_CHAIN_SETTLE", "0.3")))   # (this process's worker pools go to sleep: OpenMP / BLAS threads spin for a while after their last region)
        return _run_driver(tmp, exe, "process start to exit, beside the benchmark process (which keeps its HIP context, streams and the "
                                     "stream's 1.25 GB alive): the GPU's scheduler shares the queues of both processes")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def solver_leg(args, ctx, dev, world, rank, n_events, rate, t_start, dist, torch, np):
    import synth_solver_torch as ST
    import synth_solver as SV
    from eventcalib_amd.capi import Solver, make_allreduce_hook
    step = 5e-4                                               # MotionTimeStep, example.yaml:14
    duration = n_events / rate
    n_cp = max(4, int(duration / (50 * step)))                # EventCalibSpline.cpp:81
    n_res = int(0.9 * n_events)                               # the edge events; noise fails findCenter's gate
    if world > 1:
        # N ranks: the headline is the ONE spline of configs[2] (n_cp control points over the stream's duration, n_res
        # residuals) with its TIME cut into N shards (ecal_lm_options.distributed = 2) — total work fixed, as the metric is
        # quoted; one spline segment PER rank (per-GPU work fixed) is reported beside it
        hook = make_allreduce_hook(ctx, world) if ctx.comm_size() == 1 else None   # gloo test hook only
        out = time_shard_leg(args, ctx, dev, world, rank, dist, torch, np, n_res, n_cp, duration, hook)
        out = dict({"metric": "LM solver iterations/s", "scaling": "strong"}, **out)
        if args.scaling == "both":
            out["weak_segments_per_rank"] = segments_leg(args, ctx, dev, world, rank, dist, torch, np, n_res, n_cp, duration, hook)
        return out
    return segments_leg(args, ctx, dev, world, rank, dist, torch, np, n_res, n_cp, duration, None)


def segments_leg(args, ctx, dev, world, rank, dist, torch, np, n_res, n_cp, duration, hook):
    """One rank: THE solve of configs[2] (the headline of M2).  N ranks: every rank owns one spline segment (its own time range,
    its own solver), the intrinsics are shared by all ranks (ecal_lm_options.distributed = 1) — weak scaling, beside the
    time-sharded headline."""
    import synth_solver_torch as ST
    import synth_solver as SV
    from eventcalib_amd.capi import Solver
    # every rank owns one spline segment (its time range); intrinsics are shared by all ranks
    t0 = 5.0 + rank * (duration + 1.0)
    t1 = t0 + duration
    prob, x_seg = ST.make_problem(n_res, n_cp, t0, t1, seed=777 + rank, device=dev, round_pixels=True)
    rngp = np.random.default_rng(99)                           # same perturbation of the shared intrinsics everywhere
    intr0 = x_seg[:9].copy()
    intr0[:4] *= 1 + 0.01 * rngp.uniform(-1, 1, 4)
    intr0[4:9] += 0.01 * rngp.uniform(-1, 1, 5)
    # N > 1: every rank keeps its OWN segment in its own solver (distributed segments, include/ecal.h): what crosses
    # xGMI per evaluation is the 91-double head, per linear solve 101 + N doubles, per step 4 — never the control points
    x0 = np.concatenate([intr0, x_seg[9:]])
    combined = None
    if world > 1 and os.environ.get("ECAL_BENCH_SOLVER_CHECK") and rank == 0:
        # test hook: the same problem as ONE solver over all segments, solved by rank 0 alone — the reference result
        parts = [ST.make_problem(n_res, n_cp, 5.0 + r * (duration + 1.0), 5.0 + r * (duration + 1.0) + duration, seed=777 + r,
                                 device=dev, round_pixels=True) for r in range(world)]
        combined = (dict(parts[0][0], seg_cp_off=(np.arange(world + 1) * n_cp).astype(np.uint32),
                         knots=np.concatenate([p[0]["knots"] for p in parts]),
                         obs=np.concatenate([p[0]["obs"] for p in parts]), time=np.concatenate([p[0]["time"] for p in parts]),
                         lm_id=np.concatenate([p[0]["lm_id"] for p in parts]),
                         seg_id=np.concatenate([np.full(n_res, r, np.uint32) for r in range(world)])),
                    np.concatenate([intr0] + [p[1][9:9 + 4 * n_cp] for p in parts] + [p[1][9 + 4 * n_cp:] for p in parts]))
    solver = Solver(ctx, prob)
    del prob
    opt = solver.default_options()
    if world > 1:
        if hook is not None:
            opt.allreduce = hook
        else:   # the library's RCCL communicator, asked for explicitly (a NULL all-reduce is a rank-local solve)
            opt.allreduce, opt.allreduce_user = ctx.comm_allreduce_fn()
        opt.distributed, opt.rank, opt.world_size = 1, rank, world
    # warm-up: one whole solve of the timed length, back to back with the timed ones (an LM run is 30 ms: after any pause the
    # first solve runs on clocks that are still ramping — 275 against 295 - 300 iterations/s, profiles/r04_notes.md)
    opt.max_num_iterations = args.solver_iters
    solver.solve(x0, opt)
    # timed: three consecutive solves, the median counts (all three are in the line)
    runs = []
    for _ in range(3):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        tb = time.perf_counter()
        x, summ = solver.solve(x0, opt)
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - tb
        if world > 1:
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        runs.append(el)
    el = sorted(runs)[1]
    # kernel-level timing of one Jacobian evaluation and one cost evaluation (HIP events on the launch stream)
    st = torch.cuda.current_stream(dev)
    d_x = torch.as_tensor(x0, device=dev)
    d_acc = torch.empty(solver.n_normal, dtype=torch.float64, device=dev)
    reps = 5
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    solver.evaluate_dev(d_x.data_ptr(), 1, d_acc.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize(dev)
    ev[0].record(st)
    for _ in range(reps):
        solver.evaluate_dev(d_x.data_ptr(), 1, d_acc.data_ptr(), st.cuda_stream)
    ev[1].record(st)
    for _ in range(reps):
        solver.evaluate_dev(d_x.data_ptr(), 0, d_acc.data_ptr(), st.cuda_stream)
    ev[2].record(st)
    torch.cuda.synchronize(dev)
    jac_ms = ev[0].elapsed_time(ev[1]) / reps
    cost_ms = ev[1].elapsed_time(ev[2]) / reps
    iters = int(summ.iterations)
    res_total = n_res * world
    # the same solve with the evaluation fetched, unpacked and factorised AFTER the kernel (the form before round 4)
    plain_solve = None
    if world == 1:
        os.environ["ECAL_FORCE"] = "solver_no_stream"
        ctx.reload_env()
        try:
            solver.solve(x0, opt)
            pr = []
            for _ in range(3):
                torch.cuda.synchronize(dev)
                tb = time.perf_counter()
                xp, sp = solver.solve(x0, opt)
                torch.cuda.synchronize(dev)
                pr.append(time.perf_counter() - tb)
            plain_solve = {"value": round(int(sp.iterations) / sorted(pr)[1], 3), "unit": "iterations/s", "seconds": [round(r, 4) for r in pr],
                           "final_cost_rel_diff_vs_streamed": float(abs(sp.final_cost / summ.final_cost - 1)),
                           "note": "ECAL_FORCE=solver_no_stream: the host fetches, unpacks and factorises after the kernel has finished"}
        finally:
            del os.environ["ECAL_FORCE"]
            ctx.reload_env()
    # SURVEY 8(d)'s ALGORITHMIC count per residual and Jacobian evaluation: ~0.7 kflop residual + analytic gradient, 561 FMA
    # for the upper J^T J, 33 FMA for J^T r = 1.9 kflop (the kernel executes ~2.1 kflop: 45 matrix-core products of 4 x 4 x 4 x 4
    # blocks per 16 rows = 1440 flop per row for the 36 padded columns' tile pairs, diagonal tiles in full)
    FLOP_JAC = 1900.0
    out = {
        "metric": "LM solver iterations/s", "value": round(iters / el, 3), "unit": "iterations/s",
        "iterations": iters, "seconds": round(el, 4), "seconds_of_the_three_timed_solves": [round(r, 4) for r in runs],
        "seconds_evaluate": round(float(summ.seconds_evaluate), 4),
        "seconds_linear_solve_host": round(float(summ.seconds_linear_solve), 4), "successful_steps": int(summ.successful_steps),
        "jacobian_evaluations": int(summ.jacobian_evaluations), "cost_evaluations": int(summ.cost_evaluations),
        "initial_cost": float(summ.initial_cost), "final_cost": float(summ.final_cost),
        "residuals": res_total, "control_points": int(solver.n_cp) * world, "unknowns": int(9 + 6 * solver.n_cp * world),
        "intrinsics_rel_err_after": float(np.abs(x[:4] / SV.GT_INTR[:4] - 1).max()),
        "kernel_ms": {"normal_equations": round(jac_ms, 4), "cost_only": round(cost_ms, 4)},
        # 32 algorithmic bytes per residual and EVALUATION (SURVEY 8(d) prices an iteration at two evaluations, 64 B; this solve
        # reads the cost off the next iteration's Jacobian evaluation, so it runs about one per iteration): counted as run
        "roofline_hbm": {"algorithmic_bytes_per_residual_evaluation": 32,
                         "evaluations": int(summ.jacobian_evaluations) + int(summ.cost_evaluations), "achieved_GBs":
                         round(32.0 * n_res * (int(summ.jacobian_evaluations) + int(summ.cost_evaluations)) / el / 1e9, 2),
                         "peak_GBs": HBM_PEAK_GBS,
                         "at_the_survey_s_two_evaluations_per_iteration_GBs": round(64.0 * n_res * (iters / el) / 1e9, 2)},
        "roofline_fp64": {"kernel": "normal_eq_kernel", "flop_per_residual": FLOP_JAC,
                          "achieved_TFLOPs": round(FLOP_JAC * n_res / (jac_ms * 1e-3) / 1e12, 3), "peak_TFLOPs": 78.6,
                          "frac": round(FLOP_JAC * n_res / (jac_ms * 1e-3) / 78.6e12, 4)},
        "evaluation": "streamed: the kernel delivers the accumulation buffer group by group, the host factorises under it (design/08_solver.md)"
                      if world == 1 else "plain",
        "plain_evaluation": plain_solve,
        "sharding": "one spline segment (time range) per GPU in its own solver, shared intrinsics: 91 doubles all-reduced per "
                    "evaluation, 101 + N per linear solve, 4 per step" if world > 1 else "single GPU",
    }
    if combined is not None:
        ref = Solver(ctx, combined[0])
        o2 = ref.default_options()
        o2.max_num_iterations = args.solver_iters
        xr, sr = ref.solve(combined[1], o2)
        ref.close()
        out["check_vs_single_solver"] = {"intrinsics_rel_diff": float(np.abs(x[:9] / xr[:9] - 1).max()),
                                         "final_cost_rel_diff": float(abs(summ.final_cost / sr.final_cost - 1)),
                                         "own_control_points_abs_diff": float(np.abs(x[9:9 + 4 * n_cp] - xr[9:9 + 4 * n_cp]).max()),
                                         "iterations": [iters, int(sr.iterations)]}
    if rank == 0 and args.solver_cpu_sample > 0 and world == 1:
        import oracle_lib as O
        from concurrent.futures import ThreadPoolExecutor
        m = min(args.solver_cpu_sample, n_res)
        small, xs = ST.make_problem(m, 40, t0, t0 + 1.0, seed=5, device="cpu")
        tc = time.perf_counter()
        O.solver_evaluate(small, xs, want_H=False)
        cel = time.perf_counter() - tc
        # "cpu_ref LM with dual-number Jacobian, T threads" — T = hardware threads - 2 as the reference's workers
        # (Ceres' own threading splits the residual blocks the same way); every thread evaluates one sample of m residuals
        T = max(1, usable_cpus())      # (more threads than the cgroup lets run only queue behind each other)
        with ThreadPoolExecutor(T) as ex:
            tc = time.perf_counter()
            list(ex.map(lambda _: O.solver_evaluate(small, xs, want_H=False), range(T)))
            mel = time.perf_counter() - tc
        out["cpu_baseline"] = {"value": round(T * m / mel, 1), "unit": "residual Jacobian evaluations/s", "cores": T,
                               "kind": "port", "sample": "%d threads x %d residuals, dual-number Jacobian + gradient accumulation, "
                               "%.1f s" % (T, m, mel),
                               "implied_iterations_per_s_on_this_problem": round(T * m / mel / (2 * n_res), 6),
                               "host_cpus": os.cpu_count(), "host_cpu_quota": usable_cpus(),
                               "single_thread": {"value": round(m / cel, 1), "unit": "residual Jacobian evaluations/s", "cores": 1,
                                                 "sample": "%d residuals, %.1f s (1 of %d hardware threads)" % (m, cel, os.cpu_count() or 1)}}
    solver.close()
    return out

def host_threads_of_the_last_solve(ctx, solver, world, dist, torch, dev):
    """Threads the solver's host half used on this rank in its last solve (its pool's workers + the calling thread:
    ecal_debug_solver_last_solve), summed over the ranks, beside the CPUs the node's processes may use (affinity mask and cgroup
    quota: ecal_debug_host_usable_cpus) — N ranks on one node must share that quota, not take hardware_concurrency() each."""
    import ctypes
    L = ctx._L
    L.ecal_debug_solver_last_solve.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
    L.ecal_debug_host_usable_cpus.argtypes = [ctypes.POINTER(ctypes.c_int)]
    how = (ctypes.c_uint32 * 8)()
    ctx._check(L.ecal_debug_solver_last_solve(solver._h, how))
    quota = ctypes.c_int(0)
    per_rank = int(L.ecal_debug_host_usable_cpus(ctypes.byref(quota)))
    mine = int(how[3]) + 1
    total = mine
    if world > 1:
        tt = torch.tensor([float(mine)], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        total = int(tt.item())
    return {"this_rank": mine, "all_ranks": total, "node_cpu_quota": int(quota.value), "cpus_per_rank": per_rank,
            "local_world_size": int(os.environ.get("LOCAL_WORLD_SIZE", "1"))}


def time_shard_leg(args, ctx, dev, world, rank, dist, torch, np, n_res, n_cp, duration, hook):
    """THE spline of configs[2] (n_cp control points over the stream's duration, n_res residuals — the same
    problem whatever the number of ranks) whose residuals are cut by time into one range per rank
    (ecal_lm_options.distributed = 2): per Jacobian evaluation the 91-double head + 612 doubles per cut are all-reduced, per
    linear solve 1082 doubles per rank, nothing proportional to the control points until the solution is put together at
    the end.  Every rank generates ITS time range of the same problem."""
    import synth_solver_torch as ST
    from eventcalib_amd import capi
    from eventcalib_amd.capi import Solver
    N, C = n_res, n_cp
    ta, tb = 5.0, 5.0 + duration
    knots = ST.uniform_knots(C, ta, tb)
    cuts = np.concatenate([[-np.inf], capi.time_shard_cuts(knots, C, world), [np.inf]])
    lo_t, hi_t = cuts[rank], cuts[rank + 1]
    k_lo = 0 if rank == 0 else int((lo_t - ta) / (tb - ta) * N) - 2
    k_hi = N if rank == world - 1 else int((hi_t - ta) / (tb - ta) * N) + 3
    prob, x_gt = ST.make_problem(N, C, ta, tb, seed=4242, device=dev, round_pixels=True, k_range=(k_lo, k_hi))
    keep = (prob["time"] >= lo_t) & (prob["time"] < hi_t)
    prob = dict(prob, obs=prob["obs"][keep], time=prob["time"][keep], lm_id=prob["lm_id"][keep])
    mine = int(keep.sum())
    rngp = np.random.default_rng(99)
    x0 = x_gt.copy()
    x0[:4] *= 1 + 0.01 * rngp.uniform(-1, 1, 4)
    x0[4:9] += 0.01 * rngp.uniform(-1, 1, 5)
    solver = Solver(ctx, prob)
    del prob
    opt = solver.default_options()
    if hook is not None:
        opt.allreduce = hook
    else:
        opt.allreduce, opt.allreduce_user = ctx.comm_allreduce_fn()
    opt.distributed, opt.rank, opt.world_size = 2, rank, world
    opt.max_num_iterations = 2
    solver.solve(x0, opt)                      # warm-up
    opt.max_num_iterations = args.solver_iters
    dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    x, summ = solver.solve(x0, opt)
    torch.cuda.synchronize(dev)
    dist.barrier()
    el = time.perf_counter() - t0
    tt = torch.tensor([el, float(mine)], dtype=torch.float64, device=dev)
    tsum = tt.clone()
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    el = float(tt[0].item())
    out = {"value": round(int(summ.iterations) / el, 3), "unit": "iterations/s", "iterations": int(summ.iterations),
           "residuals": int(tsum[1].item()), "control_points": C, "unknowns": 9 + 6 * C, "seconds": round(el, 4),
           "initial_cost": float(summ.initial_cost), "final_cost": float(summ.final_cost),
           "intrinsics_rel_err_after": float(np.abs(x[:4] / x_gt[:4] - 1).max()),
           "allreduce_doubles": {"per_jacobian_evaluation": 91 + 612 * (world - 1), "per_linear_solve": 1082 * world, "per_step": 4,
                                 "once_at_the_end": int(9 + 7 * C)},
           "residuals_per_gpu_rank0": mine,
           "evaluation": "plain (a rank owns ONE interior of the partition, factorised behind the separators' all-reduce: the streamed "
                         "evaluation, which factorises under the kernel, is the single-rank path)",
           "host_threads": host_threads_of_the_last_solve(ctx, solver, world, dist, torch, dev),
           "sharding": "ONE spline (the single-GPU problem), residuals cut by time at %d knots (3-control-point separators)" % (world - 1)}
    if os.environ.get("ECAL_BENCH_SOLVER_CHECK") and rank == 0:
        # test hook: the same problem in ONE solver, solved by rank 0 alone
        full, _ = ST.make_problem(N, C, ta, tb, seed=4242, device=dev, round_pixels=True)
        ref = Solver(ctx, full)
        o2 = ref.default_options()
        o2.max_num_iterations = args.solver_iters
        xr, sr = ref.solve(x0, o2)
        ref.close()
        assert full["time"].shape[0] == out["residuals"], (full["time"].shape[0], out["residuals"])
        out["check_vs_single_solver"] = {"intrinsics_rel_diff": float(np.abs(x[:9] / xr[:9] - 1).max()),
                                         "control_points_abs_diff": float(np.abs(x[9:] - xr[9:]).max()),
                                         "final_cost_rel_diff": float(abs(summ.final_cost / sr.final_cost - 1)),
                                         "iterations": [int(summ.iterations), int(sr.iterations)]}
    solver.close()
    return out


def ingest_leg(args, ctx, dev, world, rank, dist, torch, np, rate):
    """configs[4]: the stream starts in (pinned) HOST memory; chunks are uploaded with hipMemcpyAsync on a copy stream
    while the previous chunk is detected (ecal_detect_stream_tiled) — PCIe-inclusive events/s, reported beside M1.
    N GPUs: the events are split into N time ranges, one per rank (own pinned buffer, own PCIe link, no collective);
    the rate is all events over the slowest rank's time."""
    import synth_stream as SS
    from eventcalib_amd import capi
    n = args.ingest_events // world                 # this rank's time range
    t_begin = 5.0 + rank * (n / rate)
    host = torch.empty(n * 25, dtype=torch.uint8, pin_memory=True)
    SS.CAMERA = "fisheye"       # configs[4]: the fisheye lens (Kannala-Brandt k = 0.05, -0.01, 0.002, 0: SURVEY 8d)
    try:
        host.copy_(SS.make_stream(n, rate=rate, t_start=t_begin, seed=4242 + rank, device=dev))
    finally:
        SS.CAMERA = "pinhole"
    torch.cuda.synchronize(dev)
    S = int(np.floor((n - 1) / rate / 1.5e-3)) + 1
    res = {}
    for wpc in (2048, S + 1):                       # double-buffered chunks vs one chunk (upload, then detect)
        capi.detect_stream_tiled(ctx, host.data_ptr(), n, t_begin, 1.5e-3, wpc, S + 8, want_features=False)   # warm-up (allocations)
        if world > 1:
            dist.barrier()
        tb = time.perf_counter()
        info, found, _, st = capi.detect_stream_tiled(ctx, host.data_ptr(), n, t_begin, 1.5e-3, wpc, S + 8, want_features=False)
        el = time.perf_counter() - tb
        nf = int((found != 0).sum())
        if world > 1:
            tt = torch.tensor([el, -float(nf)], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt[0].item())
        res[wpc] = (el, st["chunks"], nf)
    (e2, c2, f2), (e1, c1, f1) = res[2048], res[S + 1]
    total = n * world
    return {"metric": "Mevents/s including the PCIe upload", "value": round(total / e2 / 1e6, 1), "unit": "Mevents/s", "events": total,
            "events_per_gpu": n, "windows_per_gpu": S, "chunks_per_gpu": c2, "seconds": round(e2, 4), "grids_found_rank0": f2,
            "single_chunk_seconds": round(e1, 4), "single_chunk_Mevents_per_s": round(total / e1 / 1e6, 1),
            "pcie_floor_seconds_at_57GBs": round(n * 25 / 57e9, 4), "camera": "fisheye (Kannala-Brandt k = 0.05, -0.01, 0.002, 0)",
            "note": "stages: window bounds + slicing + DBSCAN + candidates + grid ordering per chunk of 2048 windows; the copy of "
                    "chunk k+1 overlaps the kernels of chunk k; never part of `value`"}


def calib_leg(args, ctx, dev, world, rank, dist, torch, np):
    """configs[3]: the init calibration (cv::calibrateCamera's role, EventCalibIni.cpp:198-199) on 64 views sharded
    over the GPUs; per-view J^T J / J^T r blocks on the GPU, Schur-reduced records summed by the all-reduce."""
    import synth_calib as SC
    from eventcalib_amd import capi
    V = args.calib_views
    obj, img, rv, tv = SC.make_views(V, 0, seed=2024, noise_px=0.1)      # every rank builds the same 64 views ...
    lo, hi = (V * rank) // world, (V * (rank + 1)) // world              # ... and keeps its shard
    mine = img[lo:hi]
    hook = None
    if world > 1:   # gloo test hook, or the library's RCCL communicator asked for explicitly
        hook = capi.make_allreduce_hook(ctx, world) if ctx.comm_size() == 1 else ctx.comm_allreduce_fn()
    capi.calibrate_views(ctx, obj, mine, SC.WIDTH, SC.HEIGHT, 0, SC.FLAGS_EXAMPLE, 1.0, allreduce=hook)   # warm-up
    reps = 5
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    tb = time.perf_counter()
    for _ in range(reps):
        res = capi.calibrate_views(ctx, obj, mine, SC.WIDTH, SC.HEIGHT, 0, SC.FLAGS_EXAMPLE, 1.0, allreduce=hook)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - tb
    if world > 1:
        tt = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    # kernel-level: the per-view block kernel alone (HIP events on the launch stream)
    st = torch.cuda.current_stream(dev)
    nv = hi - lo
    d_obj = torch.as_tensor(obj, device=dev)
    d_img = torch.as_tensor(np.ascontiguousarray(mine), device=dev)
    d_intr = torch.as_tensor(res["intr"], device=dev)
    d_view = torch.as_tensor(np.ascontiguousarray(np.concatenate([res["rvecs"], res["tvecs"]], 1)), device=dev)
    d_blocks = torch.empty(max(nv, 1), capi.CALIB_BLOCK_DOUBLES, dtype=torch.float64, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n_launch = 200
    for k in range(n_launch + 10):
        if k == 10:
            e0.record(st)
        capi.calib_view_blocks_dev(ctx, d_obj.data_ptr(), obj.shape[0], d_img.data_ptr(), nv, 0, SC.FLAGS_EXAMPLE, 1.0,
                                   d_intr.data_ptr(), d_view.data_ptr(), 1, d_blocks.data_ptr(), st.cuda_stream)
    e1.record(st)
    torch.cuda.synchronize(dev)
    blk_us = e0.elapsed_time(e1) / n_launch * 1e3
    out = {
        "metric": "init calibrations/s (64 views)", "value": round(reps / el, 3), "unit": "calibrations/s",
        "ms_per_calibration": round(el / reps * 1e3, 3), "lm_iterations": int(res["iterations"]),
        "lm_iterations_per_s": round(res["iterations"] * reps / el, 1),
        "jacobian_evaluations": int(res["jacobian_evaluations"]), "views_total": V, "views_per_gpu": nv,
        "rms_px": float(res["rms"]), "fx_rel_err": float(abs(res["intr"][0] / SC.GT_PINHOLE[0] - 1)),
        "view_blocks_kernel_us": round(blk_us, 2),
        "allreduce_doubles_per_evaluation": 170,
        "scaling": "strong (the 64 views of configs[3] are split over the GPUs)",
        "note": "model pinhole + radtan with example.yaml's flags; every LM iteration = block kernel + Schur kernel + "
                "170-double reduction (+ RCCL all-reduce when sharded) + 12x12 host solve + back-substitution kernel",
    }
    if rank == 0 and world == 1 and args.calib_cpu_views > 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import calib_oracle as CO
        m = min(args.calib_cpu_views, V)
        tc = time.perf_counter()
        CO.calibrate(0, obj, img[:m], SC.WIDTH, SC.HEIGHT, SC.FLAGS_EXAMPLE, 1.0)
        cel = time.perf_counter() - tc
        out["cpu_baseline"] = {"value": round(1.0 / cel, 4), "unit": "calibrations/s", "cores": 1, "kind": "port",
                               "host_cpus": os.cpu_count(), "sample": "1 thread of %d; %d of the %d views, numpy restatement with finite-difference Jacobians and the "
                                         "dense (12 + 6V)^2 solve OpenCV uses, %.1f s" % (os.cpu_count() or 1, m, V, cel)}
    return out


if __name__ == "__main__":
    main()
