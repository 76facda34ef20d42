"""CPU, world_size 2 (gloo): the N > 1 paths shard and reduce correctly.

* detection shards by time range with no data-path collective: two ranks processing disjoint window ranges
  produce, put together, exactly the single-rank result (checked with the oracle as evaluator);
* the solver shards residuals by time range / spline segment and all-reduces the normal-equation
  buffer: sum of per-rank partials == whole-problem evaluation.
The GPU kernels cannot run here; the oracle stands in as the per-rank evaluator, so what is under test is
the partitioning and the reduction layout that bench.py and eventcalib_amd use."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as O
    import synth_solver as SV
    import synth_stream as SS
    try:
        # ---- detection: contiguous window ranges per rank ----
        buf = SS.make_stream(30000, device="cpu", seed=3)
        t, _, _ = SS.unpack_records(buf)
        t0, t1 = SS.tiled_windows(float(t[0]), float(t[-1]))
        S = len(t0)
        lo, hi = (S * rank) // world, (S * (rank + 1)) // world
        ev, ncl = O.detect_windows(buf.numpy(), t0[lo:hi], t1[lo:hi], 4.0, 2)
        tot = torch.tensor([ev, ncl], dtype=torch.float64)
        dist.all_reduce(tot)
        ev_all, ncl_all = O.detect_windows(buf.numpy(), t0, t1, 4.0, 2)
        assert int(tot[0]) == ev_all == 30000 and int(tot[1]) == ncl_all

        # ---- solver: two segments, one per rank, shared intrinsics; all-reduce of the buffer ----
        n_cp = 6
        prob, x = SV.make_problem(800, n_cp=n_cp, seed=11, n_segments=world, pixel_noise=0.3)
        m = prob["seg_id"] == rank
        mine = dict(prob, seg_cp_off=np.array([0, n_cp], np.uint32), knots=prob["knots"][10 * rank: 10 * rank + 10],
                    obs=prob["obs"][m], time=prob["time"][m], lm_id=prob["lm_id"][m], seg_id=None)
        xm = np.concatenate([x[:9], x[9 + 4 * n_cp * rank: 9 + 4 * n_cp * (rank + 1)],
                             x[9 + 4 * n_cp * world + 3 * n_cp * rank: 9 + 4 * n_cp * world + 3 * n_cp * (rank + 1)]])
        c, g, H = O.solver_evaluate(mine, xm)
        n = 9 + 6 * n_cp * world
        G = np.zeros(n)
        Hf = np.zeros((n, n))
        idx = np.concatenate([np.arange(9), 9 + 6 * n_cp * rank + np.arange(6 * n_cp)])
        G[idx] = g
        Hf[np.ix_(idx, idx)] = H
        pack = torch.from_numpy(np.concatenate([[c], G, Hf.ravel()]))
        dist.all_reduce(pack)
        pack = pack.numpy()
        # every rank now holds the same reduced system; intrinsics blocks are sums over ranks
        ref = torch.from_numpy(pack.copy())
        dist.broadcast(ref, 0)
        assert np.array_equal(ref.numpy(), pack)
        c_other = pack[0] - c
        assert c_other > 0 and abs(pack[1:10] - g[:9]).max() > 0       # the other rank contributed
        # cross-segment control-point blocks stay zero
        Hr = pack[1 + n:].reshape(n, n)
        assert np.abs(Hr[9:9 + 6 * n_cp, 9 + 6 * n_cp:]).max() == 0.0
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, "FAIL: %r" % (e,)))
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
