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
        # ---- distributed segments (ecal_lm_options.distributed): every rank eliminates its own control points; only the
        #      head (intrinsics gradient / block) and the 9 x 10 Schur sums are all-reduced.  Step == dense solve. ----
        gi_loc, Hii_loc = g[:9], H[:9, :9]
        Hcc, Hci, gc = H[9:, 9:], H[9:, :9], g[9:]
        lam = 1e-3
        Acc = Hcc + lam * np.diag(np.diag(Hcc))
        W = np.linalg.solve(Acc, np.concatenate([Hci, gc[:, None]], axis=1))          # Acc^-1 [B | g_c]
        piece = torch.from_numpy(np.concatenate([(Hci.T @ W).ravel(), gi_loc, Hii_loc.ravel()]))   # 90 + 9 + 81
        dist.all_reduce(piece)
        piece = piece.numpy()
        P, gi_all, Hii_all = piece[:90].reshape(9, 10), piece[90:99], piece[99:].reshape(9, 9)
        Sred = Hii_all + lam * np.diag(np.diag(Hii_all)) - P[:, :9]
        step_i = np.linalg.solve(Sred, -(gi_all - P[:, 9]))
        step_c = -np.linalg.solve(Acc, gc + Hci @ step_i)                             # own control points only
        Hs = Hr + lam * np.diag(np.diag(Hr))
        full = np.linalg.solve(Hs, -pack[1:1 + n])
        assert np.allclose(step_i, full[:9], rtol=1e-7, atol=1e-12)
        own = slice(9 + 6 * n_cp * rank, 9 + 6 * n_cp * (rank + 1))
        assert np.allclose(step_c, full[own], rtol=1e-6, atol=1e-10)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, "FAIL: %r" % (e,)))
    finally:
        dist.destroy_process_group()


def _calib_worker(rank, world, port, q):
    """Init calibration, views sharded over ranks: the all-reduced 170-double Schur records give every rank the
    step the dense (12 + 6V)^2 system gives (what cv::calibrateCamera solves)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
    import calib_oracle as CO
    import synth_calib as SC
    try:
        V, lam = 6, 1e-3
        obj, img, rv, tv = SC.make_views(V, 0, seed=9, noise_px=0.3)
        flags, aspect = SC.FLAGS_EXAMPLE, 1.0
        p = np.concatenate([SC.GT_PINHOLE * (1 + 1e-3)] + [np.concatenate([rv[v], tv[v]]) for v in range(V)])
        lo, hi = (V * rank) // world, (V * (rank + 1)) // world
        rec = torch.from_numpy(CO.reduced_record(0, flags, aspect, p, obj, img, range(lo, hi), lam))
        assert rec.numel() == 170
        dist.all_reduce(rec)
        rec = rec.numpy()
        assert rec[169] == V * obj.shape[0]
        free = np.flatnonzero(CO.free_mask(0, flags))
        A = rec[:144].reshape(12, 12)[np.ix_(free, free)] + lam * np.diag(rec[156:168][free])
        x_red = np.linalg.solve(A, rec[144:156][free])
        # dense reference: the whole arrow system with the (1 + lambda) diagonal, as CvLevMarq::step
        fm = np.concatenate([CO.free_mask(0, flags), np.ones(6 * V)])
        J = CO.jacobian_fd(0, flags, aspect, p, obj, img) * fm
        r = CO.residuals(0, flags, aspect, p, obj, img)
        idx = np.flatnonzero(fm)
        H = (J.T @ J)[np.ix_(idx, idx)]
        H[np.diag_indices_from(H)] *= 1 + lam
        x_full = np.linalg.solve(H, (J.T @ r)[idx])
        assert np.allclose(x_red, x_full[:len(free)], rtol=1e-6, atol=1e-9), (x_red, x_full[:len(free)])
        assert abs(rec[168] - r @ r) < 1e-9
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, "FAIL: %r" % (e,)))
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo_calibration_views():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30100 + (os.getpid() % 500)
    procs = [ctx.Process(target=_calib_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


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


def _handover_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
        from eventcalib_amd.adaptive import DistHandover
        # the chain in time of `bench.py --gpus N`'s sharded shared-map search: rank r holds pieces later in time than rank r + 1
        ho = DistHandover(rank + 1 if rank + 1 < world else None, rank - 1 if rank > 0 else None, tag=7)
        if rank + 1 == world:
            got = (0, 0.0, [0.0] * 64)              # holds the run's first piece: the library never asks it to receive
        else:
            polls = 0
            got = ho.recv(False)
            while got is None and polls < 200000:   # between passes: "not there yet" until the predecessor is done
                polls += 1
                got = ho.recv(False)
            if got is None:
                got = ho.recv(True)
            assert ho.recv(False) == got and ho.recv(True) == got    # kept: a repeated call finds it again
        has, t, dirs = got
        assert len(dirs) == 64
        mine = (1, t + 10.0 + rank, [float(rank)] * 18 + [0.0] * 46) if rank % 2 == 0 else got   # odd ranks: no keyframe of their own
        ho.send(*mine)
        ho.send(1, -1.0, [9.0] * 64)                 # a second send (a call repeated with more capacity) is a no-op
        q.put((rank, got[0], got[1], list(got[2][:2]), None))
    except Exception as e:  # noqa: BLE001
        q.put((rank, None, None, None, repr(e)))
    finally:
        dist.destroy_process_group()


def test_world_size_3_gloo_keyframe_frame_handover():
    """DistHandover (eventcalib_amd/adaptive.py), the transport of ecal_detect_keyframes_sharded's frame hand-over: a chain of
    world - 1 point-to-point messages from the rank that is earliest in time to the latest; polling receive, the frame kept for
    repeated calls, one send per rank; a rank without a keyframe of its own passes on what it received."""
    world, port = 3, 29771 + os.getpid() % 100
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    ps = [ctxm.Process(target=_handover_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(60)
    assert all(r[4] is None for r in res), res
    # rank 2 (earliest): nothing before it, sends (1, 12, [2, 2, ..]); rank 1 (odd) passes it on; rank 0 receives rank 2's frame
    assert res[2][1:4] == (0, 0.0, [0.0, 0.0])
    assert res[1][1:4] == (1, 12.0, [2.0, 2.0])
    assert res[0][1:4] == (1, 12.0, [2.0, 2.0])
