"""GPU: the C++ shims (eventcalib_amd/csrc/host) compile against include/ecal.h + libecal.so and
behave like the reference classes they replace."""
import os
import subprocess

import pytest

import synth_stream as SS

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_shims(tmp_path):
    exe = str(tmp_path / "test_shims")
    lib_dir = os.path.join(ROOT, "eventcalib_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_shims.cpp"),
                           "-L" + lib_dir, "-lecal", "-Wl,-rpath," + lib_dir, "-lpthread"])
    binf = str(tmp_path / "events.bin")
    SS.make_stream(60000, rate=2.0e6, device="cpu").numpy().tofile(binf)
    import numpy as np
    import synth_rectify as SR
    times = 5.0 + np.arange(0, 400) * 1e-4
    posef = str(tmp_path / "poses.bin")
    np.concatenate([times[:, None], SR.poses_cw(times)], axis=1).astype(np.float64).tofile(posef)
    out = subprocess.run([exe, binf, posef], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "shims ok" in out.stdout and "rectify:" in out.stdout
    # the Python mirror of the adaptive windowing (eventcalib_amd/adaptive.py) must select the same keyframes
    import torch
    import eventcalib_amd
    from eventcalib_amd.adaptive import detect_keyframes
    from eventcalib_amd.pipeline import DetectPipeline
    buf = torch.from_numpy(np.fromfile(binf, dtype=np.uint8))
    t, _, _ = SS.unpack_records(buf)
    ctx = eventcalib_amd.Context(0)
    kf = detect_keyframes(DetectPipeline(ctx), buf.cuda(), 5e-4, 4000, 4, float(t[0]), float(t[-1]))
    rows = [ln.split()[1:] for ln in out.stdout.splitlines() if ln.startswith("kf ")]
    assert len(rows) == len(kf["time"]) >= 2
    for r, d, n, f in zip(rows, kf["duration"], kf["events_num"], kf["features"]):
        assert abs(float(r[0]) - d[0]) < 1e-9 and abs(float(r[1]) - d[1]) < 1e-9 and int(r[2]) == n
        assert abs(float(r[3]) - f[0, 0]) < 1e-6 and abs(float(r[4]) - f[35, 2]) < 1e-6
    ctx.close()
