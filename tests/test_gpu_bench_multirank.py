"""GPU: bench.py's N > 1 code paths (time-range sharding of detection, segment-per-rank solver with the all-reduce
hook, views-per-rank init calibration) on a one-GPU box: two ranks share device 0 and the collectives run over gloo
(ECAL_BENCH_SINGLE_DEVICE / ECAL_BENCH_BACKEND, test hooks the driver never sets).  The 8-GPU RCCL run is the
driver's; here the plumbing and the sharded results are checked."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(nproc, extra, launcher=True):
    env = dict(os.environ, ECAL_BENCH_SINGLE_DEVICE="1", ECAL_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0",
               ECAL_BENCH_SOLVER_CHECK="1")
    args = ["--gpus", str(nproc), "--steps", "2", "--warmup", "1", "--events", "2000000", "--cpu-sample", "0",
            "--solver-iters", "3", "--solver-cpu-sample", "0", "--p2-pieces", "0", "--no-h2d", "--calib-cpu-views", "0", "--ingest-events", "1000000", "--e2e-events", "0"] + extra
    if nproc == 1 or not launcher:
        # exactly what the driver may type: `python bench.py --gpus N ...` — for N > 1 bench.py starts its own ranks under
        # torch.distributed.run as child processes (before it touches the GPU) and relays rank 0's JSON line
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
               "--master-addr", "127.0.0.1", "--master-port", str(29650 + os.getpid() % 200), os.path.join(ROOT, "bench.py")] + args
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_gpus_flag_without_a_launcher_starts_the_ranks_itself():
    two = _bench(2, ["--solver-iters", "0", "--calib-views", "0"], launcher=False)
    assert two["n_gpus"] == 2 and two["config"]["events_total"] == 2000000 and two["value"] > 0 and two["failed_legs"] == []


def test_two_ranks_match_one_rank():
    one = _bench(1, ["--p2-pieces", "40"])
    two = _bench(2, ["--p2-pieces", "40"])
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and one["failed_legs"] == [] and two["failed_legs"] == []
    # the HEADLINE is strong scaling (BASELINE.json's metric: "on 50M-event stream @1/2/4/8 GPU"): ONE 2 M-event stream cut into
    # two time ranges of whole windows — together the ranks' windows cover every event of it exactly once; value = the whole
    # stream's events over the slowest rank's time (the same definition as with one rank)
    for r in (one, two):
        assert r["scaling"] == "strong" and r["config"]["events_total"] == 2000000
        assert r["config"]["events_covered_by_the_ranks_windows"] == 2000000
        assert abs(r["value"] - 2000000 * r["steps"] / (r["ms_per_step"] * 1e-3 * r["steps"]) / 1e6) < 1e-3 * r["value"]
        assert "HBM-resident" in r["config"]["workload"] and "upload excluded" in r["config"]["workload"]
    assert two["config"]["windows_total"] == one["config"]["windows_total"] == one["config"]["windows_per_gpu"]
    assert two["config"]["windows_per_gpu"] == one["config"]["windows_per_gpu"] // 2 and two["config"]["events_per_gpu"] < 1_010_000
    assert "strong_scaling" not in two and "weak_scaling" not in one
    # ... weak scaling (one 2 M-event stream per rank) is reported beside it
    ws = two["weak_scaling"]
    assert ws["scaling"] == "weak" and ws["events_per_gpu"] == 2000000 and ws["value"] > 0
    # the non-scaling part of a pass is measured (what caps strong scaling at N = 8)
    assert one["pass_ms_fixed"]["ms"] < 0.5 * one["ms_per_step"] + 0.2
    # ... and the adaptive-window search of that stream with its pieces cut over the ranks: the keyframes and windows of the
    # single-rank search (own-piece gate, same piece count)
    p1 = [p for p in one["policy_p2"] if p.get("driver") == "device" and p.get("gate") == "own piece"][0]
    p2 = two["policy_p2_sharded"]["own_piece_gate"]
    assert p2["pieces"] == p1["pieces"] and p2["pieces_per_gpu"] <= (p1["pieces"] + 1) // 2
    assert p2["keyframes"] == p1["keyframes"] > 0 and p2["windows_evaluated"] == p1["windows_evaluated"]
    # ... and with the reference's gate — the single-GPU front ends' default — through the frame hand-over between the ranks
    # (ecal_detect_keyframes_sharded): the very keyframes of the single-rank shared-map search
    s1 = [p for p in one["policy_p2"] if p.get("driver") == "device" and p.get("gate", "").startswith("shared map")][0]
    s2 = two["policy_p2_sharded"]
    assert s2["gate"].startswith("shared map") and s2["pieces"] == s1["pieces"]
    assert s2["keyframes"] == s1["keyframes"] > 0 and s2["windows_evaluated"] == s1["windows_evaluated"]
    assert s2["keyframe_times_sha1_16"] == s1["keyframe_times_sha1_16"] and s2["keyframe_times_sha1_16"] != p2["keyframe_times_sha1_16"]
    # the event -> point map: the line says whether the timed pass wrote it, and carries both figures
    for r in (one, two):
        assert r["config"]["event_point"] is False and r["event_point_map"]["timed_pass_writes_the_map"] is False
        assert r["event_point_map"]["ms_per_step_with_the_map"] > 0 and r["event_point_map"]["ms_per_step_without_the_map"] > 0
    # the sharded init calibration lands on the single-rank answer (same 64 views, Schur records summed over ranks)
    c1, c2 = one["init_calibration"], two["init_calibration"]
    assert c1["views_per_gpu"] == 64 and c2["views_per_gpu"] == 32
    # same minimum; the stop test (relative parameter change below DBL_EPSILON) sits on rounding, and the sharded sum
    # adds the Schur records in a different order, so the count may differ by an iteration
    assert abs(c1["rms_px"] - c2["rms_px"]) < 1e-9 and abs(c1["lm_iterations"] - c2["lm_iterations"]) <= 1
    assert c1["fx_rel_err"] < 2e-3 and c2["fx_rel_err"] < 2e-3
    # M2 under two ranks: the headline is the SAME spline as with one rank, its time cut into two shards (distributed = 2)
    s1, s2 = one["solver"], two["solver"]
    assert s2["scaling"] == "strong" and s2["residuals"] == s1["residuals"] and s2["control_points"] == s1["control_points"]
    assert s2["unknowns"] == s1["unknowns"] and s2["value"] > 0
    assert s2["intrinsics_rel_err_after"] < 8e-3 and s1["intrinsics_rel_err_after"] < 8e-3
    assert s2["final_cost"] < s2["initial_cost"]
    tc = s2["check_vs_single_solver"]
    assert tc["iterations"][0] == tc["iterations"][1]
    assert tc["intrinsics_rel_diff"] < 1e-8 and tc["final_cost_rel_diff"] < 1e-9 and tc["control_points_abs_diff"] < 1e-6
    assert s2["allreduce_doubles"]["per_jacobian_evaluation"] == 91 + 612
    # two ranks on one box share the CPUs the node's processes may use (affinity mask and cgroup quota over LOCAL_WORLD_SIZE): the
    # solver's worker pools together stay within it — not hardware_concurrency() threads per rank
    ht = s2["host_threads"]
    assert ht["local_world_size"] == 2 and ht["cpus_per_rank"] == max(1, ht["node_cpu_quota"] // 2)
    assert ht["all_ranks"] <= max(2, ht["node_cpu_quota"]) and ht["this_rank"] <= ht["cpus_per_rank"]
    # ... beside it, one spline segment per rank (weak): distributed segments (each rank factorises its own band, 91 / 101 + N /
    # 4 doubles exchanged) == one solver over both segments: same iterates up to the summation order
    wk = s2["weak_segments_per_rank"]
    assert wk["residuals"] == 2 * s1["residuals"] and wk["final_cost"] < wk["initial_cost"]
    chk = wk["check_vs_single_solver"]
    assert chk["iterations"][0] == chk["iterations"][1]
    assert chk["intrinsics_rel_diff"] < 1e-8 and chk["final_cost_rel_diff"] < 1e-9 and chk["own_control_points_abs_diff"] < 1e-7
    # the ingest leg (configs[4]): the host-resident events split into one time range per rank
    assert one["ingest"]["events"] == two["ingest"]["events"] == 1000000 and two["ingest"]["events_per_gpu"] == 500000
    assert one["ingest"]["value"] > 0 and two["ingest"]["value"] > 0


def test_eight_ranks_match_one_rank():
    """The 8-way partitions of `bench.py --gpus 8` (what the driver launches on the 8-GPU node), executed here with eight ranks on
    ONE GPU over gloo: 8 time ranges of the detection stream, the keyframe search's pieces cut over 8 ranks, 64 calibration views
    8 per rank, the spline's time cut into 8 shards (distributed = 2) — each against the single-rank result.  No scaling figure is
    read off this (eight contexts share one device); it shows that the partitions and the exchange are right at N = 8."""
    one = _bench(1, ["--p2-pieces", "40"])
    eight = _bench(8, ["--p2-pieces", "40"])
    assert eight["n_gpus"] == 8 and eight["failed_legs"] == []
    assert eight["config"]["events_covered_by_the_ranks_windows"] == 2000000 == eight["config"]["events_total"]
    assert eight["config"]["windows_total"] == one["config"]["windows_total"]
    assert eight["config"]["windows_per_gpu"] <= one["config"]["windows_per_gpu"] // 8 + 1
    p1 = [p for p in one["policy_p2"] if p.get("driver") == "device" and p.get("gate") == "own piece"][0]
    p8 = eight["policy_p2_sharded"]["own_piece_gate"]
    assert p8["pieces"] == p1["pieces"] and p8["keyframes"] == p1["keyframes"] > 0 and p8["windows_evaluated"] == p1["windows_evaluated"]
    s1 = [p for p in one["policy_p2"] if p.get("driver") == "device" and p.get("gate", "").startswith("shared map")][0]
    s8 = eight["policy_p2_sharded"]
    assert s8["gate"].startswith("shared map") and s8["keyframes"] == s1["keyframes"] > 0
    assert s8["windows_evaluated"] == s1["windows_evaluated"] and s8["keyframe_times_sha1_16"] == s1["keyframe_times_sha1_16"]
    c1, c8 = one["init_calibration"], eight["init_calibration"]
    assert c8["views_per_gpu"] == 8 and abs(c1["rms_px"] - c8["rms_px"]) < 1e-9 and abs(c1["lm_iterations"] - c8["lm_iterations"]) <= 1
    s1, s8 = one["solver"], eight["solver"]
    assert s8["scaling"] == "strong" and s8["residuals"] == s1["residuals"] and s8["unknowns"] == s1["unknowns"]
    tc = s8["check_vs_single_solver"]
    assert tc["iterations"][0] == tc["iterations"][1]
    assert tc["intrinsics_rel_diff"] < 1e-8 and tc["final_cost_rel_diff"] < 1e-9 and tc["control_points_abs_diff"] < 1e-6
    assert eight["ingest"]["events_per_gpu"] == 125000
