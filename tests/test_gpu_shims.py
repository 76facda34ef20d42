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
