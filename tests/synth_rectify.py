"""Shared inputs for the rectifyFeatures tests: poses / camera of the synthetic stream (synth_stream.py)."""
import numpy as np
import torch

import synth_stream as SS

CAMERA = (SS.FX, SS.FY, SS.CX, SS.CY)
DIST = (SS.K1, SS.K2, 0.0, 0.0, SS.K3)          # cv::projectPoints order: k1 k2 p1 p2 k3


def poses_cw(times, shift=None):
    """[n,12] = Rcw row-major + tcw at `times`; `shift` [n,3] moves the camera centre (cm) to fake a bad PnP pose."""
    R_wc, C = SS.pose(torch.as_tensor(np.asarray(times, dtype=np.float64)))
    R_wc, C = R_wc.numpy(), C.numpy()
    if shift is not None:
        C = C + shift
    out = np.zeros((len(times), 12))
    for i in range(len(times)):
        Rcw = R_wc[i].T
        out[i, :9] = Rcw.reshape(-1)
        out[i, 9:] = -Rcw @ C[i]
    return out


def landmarks_f32():
    """Landmark positions as the reference holds them: cv::Point3f widened back to double (EventCalibIni.cpp:102-106,227)."""
    return SS.landmarks().numpy().astype(np.float32).astype(np.float64)
