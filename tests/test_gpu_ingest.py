"""GPU: double-buffered ingest (ecal_detect_stream_tiled: chunks uploaded with hipMemcpyAsync while the previous chunk
is processed) gives exactly what the resident pipeline gives on the same tiled windows — window verdicts, grid flags
and the ordered circles — for several chunk sizes, including one chunk and chunks of a single window."""
import numpy as np
import pytest

import synth_stream as SS

pytestmark = pytest.mark.gpu


def test_streamed_equals_resident():
    import torch
    import eventcalib_amd
    from eventcalib_amd import capi
    from eventcalib_amd.pipeline import DetectPipeline
    n, rate = 800_000, 4.0e6                     # dense enough for complete grids in 1.5 ms windows
    buf = SS.make_stream(n, rate=rate, t_start=5.0, device="cpu", seed=9)
    host = buf.pin_memory()
    t0, t1 = SS.tiled_windows(5.0, 5.0 + (n - 1) / rate, 1.5e-3)
    S = len(t0)
    ctx = eventcalib_amd.Context(0)
    pipe = DetectPipeline(ctx)
    pipe.set_windows(t0, t1)
    pipe.run(buf.cuda())
    order, found = pipe.order_grid(9, 4)
    torch.cuda.synchronize()
    info_ref = pipe.win_info[:S].cpu().numpy().astype(np.uint32)
    found_ref = found.cpu().numpy().astype(np.uint32)
    base = pipe.seg_off[: 2 * S: 2].long()
    feat_ref = pipe.cand_xyr[(base[:, None] + order.long().clamp(min=0)).reshape(-1)].reshape(S, 36, 3).cpu().numpy()
    ok = (info_ref[:, 3] == 0) & (found_ref != 0)
    assert ok.sum() > S // 4
    for wpc in (S + 5, 64, 7, 1):
        info, fnd, feat, st = capi.detect_stream_tiled(ctx, host.data_ptr(), n, 5.0, 1.5e-3, wpc, S + 8)
        assert len(info) == S and st["chunks"] == -(-S // wpc) and st["bytes_uploaded"] == n * 25
        assert np.array_equal(info, info_ref) and np.array_equal(fnd, found_ref)
        assert np.array_equal(feat[ok], feat_ref[ok]) and np.isnan(feat[~ok]).all()
    # error paths: too few result rows, no events
    with pytest.raises(capi.EcalError):
        capi.detect_stream_tiled(ctx, host.data_ptr(), n, 5.0, 1.5e-3, 16, S - 1)
    info, fnd, feat, st = capi.detect_stream_tiled(ctx, host.data_ptr(), 0, 5.0, 1.5e-3, 16, 8)
    assert len(info) == 0
    ctx.close()
