"""GPU: ecal_stream_create_from_file — the reference's reading loop (eventCameraCalib.cpp:154-163: keep timeStamp >= StartTime,
stop at the first timeStamp >= EndTime when an end is set) straight from the .bin file into HBM, chunked reads overlapped with
the upload — against the same rule applied to the file's records in numpy, across chunk boundaries (2^20 records a chunk),
for a file that is not in time order (the stream comes out in the multimap's order) and for a missing file."""
import ctypes
import os
import tempfile

import numpy as np
import pytest

import synth_stream as SS

pytestmark = pytest.mark.gpu


def _from_file(ctx, torch, path, start, has_end, end):
    L = ctx._L
    vp = ctypes.c_void_p
    L.ecal_stream_create_from_file.argtypes = [vp, ctypes.c_char_p, ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.POINTER(vp)]
    L.ecal_stream_create_from_file.restype = ctypes.c_int
    L.ecal_stream_size.argtypes = [vp]
    L.ecal_stream_size.restype = ctypes.c_uint64
    L.ecal_stream_data.argtypes = [vp]
    L.ecal_stream_data.restype = vp
    L.ecal_stream_times.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    L.ecal_stream_times.restype = ctypes.c_int
    L.ecal_stream_destroy.argtypes = [vp]
    L.ecal_stream_destroy.restype = None
    h = vp()
    rc = L.ecal_stream_create_from_file(ctx._h, path.encode(), float(start), int(has_end), float(end), ctypes.byref(h))
    if rc != 0:
        return rc, None, None
    n = int(L.ecal_stream_size(h))
    out = torch.empty(max(n, 1) * 25, dtype=torch.uint8, device="cuda")
    if n:
        ctx._check(L.ecal_copy_dev(ctx._h, out.data_ptr(), L.ecal_stream_data(h), n * 25, torch.cuda.current_stream().cuda_stream, 1))
    t0, t1 = ctypes.c_double(), ctypes.c_double()
    assert L.ecal_stream_times(h, ctypes.byref(t0), ctypes.byref(t1)) == 0
    L.ecal_stream_destroy(h)
    return 0, out[: n * 25].cpu().numpy().reshape(n, 25), (t0.value, t1.value)


def _rule(rec, start, has_end, end):
    t = rec[:, :8].copy().view(np.float64).ravel()
    stop = len(t)
    if has_end:
        hit = np.flatnonzero(t >= end)
        if len(hit):
            stop = int(hit[0])
    keep = np.flatnonzero(t[:stop] >= start)
    return rec[keep]


def test_stream_from_file():
    import torch
    import eventcalib_amd
    ctx = eventcalib_amd.Context(0)
    try:
        n = 2_600_000                                     # three chunks
        buf = SS.make_stream(n, rate=1.0e6, device="cpu", seed=8)
        rec = buf.numpy().reshape(n, 25)
        t = rec[:, :8].copy().view(np.float64).ravel()
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "events.bin")
            rec.tofile(path)
            with open(path, "ab") as f:
                f.write(b"\x01" * 7)                      # a trailing partial record is ignored (EventStream reads whole records)
            for start, has_end, end in ((-1.0, 0, 0.0), (float(t[1_300_000]), 0, 0.0), (float(t[10]), 1, float(t[2_200_003])),
                                        (float(t[1_048_570]), 1, float(t[1_048_580])), (1e9, 0, 0.0), (0.0, 1, -5.0)):
                rc, got, times = _from_file(ctx, torch, path, start, has_end, end)
                want = _rule(rec, start, has_end, end)
                assert rc == 0 and got.shape == want.shape and np.array_equal(got, want), (start, has_end, end)
                if len(want):
                    assert times == (float(want[0, :8].copy().view(np.float64)[0]), float(want[-1, :8].copy().view(np.float64)[0]))
                else:
                    assert times == (0.0, 0.0)
            # not in time order: kept by the same rule in FILE order (a record below StartTime anywhere is dropped, the first
            # one at or beyond EndTime ends the reading), then the multimap's order — stable by time stamp
            rng = np.random.default_rng(3)
            perm = rng.permutation(200_000)
            shuffled = rec[:200_000][perm]
            p2 = os.path.join(tmp, "shuffled.bin")
            shuffled.tofile(p2)
            for start, has_end, end, at_least in ((float(t[20_000]), 0, 0.0, 150_000), (float(t[20_000]), 1, float(t[199_000]), 50)):
                rc, got, _ = _from_file(ctx, torch, p2, start, has_end, end)
                kept = _rule(shuffled, start, has_end, end)
                tk = kept[:, :8].copy().view(np.float64).ravel()
                want = kept[np.argsort(tk, kind="stable")]
                assert rc == 0 and at_least < len(want) < 200_000 and np.array_equal(got, want), (has_end, len(want))
            rc, _, _ = _from_file(ctx, torch, os.path.join(tmp, "missing.bin"), 0.0, 0, 0.0)
            assert rc == -1
            p3 = os.path.join(tmp, "short.bin")               # no whole record: an empty stream, as an EventStream at its end
            with open(p3, "wb") as f:
                f.write(b"\x00" * 24)
            rc, got, times = _from_file(ctx, torch, p3, 0.0, 0, 0.0)
            assert rc == 0 and got.shape == (0, 25) and times == (0.0, 0.0)
    finally:
        ctx.close()
