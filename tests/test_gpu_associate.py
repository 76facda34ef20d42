"""GPU parity: event -> residual association vs the oracle (bit-exact: same arithmetic, index outputs)."""
import numpy as np
import pytest

import oracle_lib as O
import synth_stream as SS

pytestmark = pytest.mark.gpu


def _keyframes(t_first, t_last, torch):
    """Keyframes every 4 ms with the projected ground-truth circles of the synthetic stream."""
    kt = np.arange(t_first + 2e-3, t_last, 4e-3)
    tt = torch.tensor(kt, dtype=torch.float64)
    R, C = SS.pose(tt)
    lm = SS.landmarks()
    circ = np.zeros((len(kt), 36, 3))
    for i in range(36):
        c = SS.project(lm[i][None, :].expand(len(kt), 3), R, C).numpy()
        e = SS.project((lm[i] + torch.tensor([SS.RADIUS, 0.0, 0.0]))[None, :].expand(len(kt), 3), R, C).numpy()
        circ[:, i, :2] = c
        circ[:, i, 2] = np.linalg.norm(e - c, axis=1)
    return kt, circ


def test_associate_matches_oracle():
    import torch
    import eventcalib_amd
    ctx = eventcalib_amd.Context(0)
    buf = SS.make_stream(120000, device="cpu", seed=8)
    t, _, _ = SS.unpack_records(buf)
    t_first, t_last = float(t[0]), float(t[-1])
    kt, circ = _keyframes(t_first, t_last, torch)
    keep = (np.arange(len(kt)) % 7) != 3              # drop some keyframes: events too far from any are rejected
    kt, circ = kt[keep], circ[keep]
    d_ev = buf.cuda()
    d_kt = torch.tensor(kt).cuda()
    d_ci = torch.tensor(circ).cuda()
    n = 120000
    obs = torch.empty(n, 2, dtype=torch.float64, device="cuda")
    tm = torch.empty(n, dtype=torch.float64, device="cuda")
    lm = torch.empty(n, dtype=torch.int32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    for (t_min, t_max) in ((t_first, t_last), (t_first + 0.01, t_last - 0.02)):
        ctx.associate_dev(d_ev.data_ptr(), n, d_kt.data_ptr(), d_ci.data_ptr(), len(kt), 36, t_min, t_max, 2.5e-3, 5.0,
                          obs.data_ptr(), tm.data_ptr(), lm.data_ptr(), cnt.data_ptr(), 0)
        torch.cuda.synchronize()
        m = int(cnt.item())
        oo, ot, ol = O.associate(buf.numpy(), kt, circ, t_min, t_max, 2.5e-3, 5.0)
        assert m == len(ot) and m > 10000
        assert np.array_equal(obs[:m].cpu().numpy(), oo)
        assert np.array_equal(tm[:m].cpu().numpy(), ot)
        assert np.array_equal(lm[:m].cpu().numpy().astype(np.uint32), ol)
    # no keyframes -> nothing associated
    ctx.associate_dev(d_ev.data_ptr(), n, 0, 0, 0, 36, t_first, t_last, 2.5e-3, 5.0, obs.data_ptr(), tm.data_ptr(),
                      lm.data_ptr(), cnt.data_ptr(), 0)
    torch.cuda.synchronize()
    assert int(cnt.item()) == 0
    ctx.close()
