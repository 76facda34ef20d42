"""CPU suite: the YAML-subset settings reader of the C++ shims (eventcalib_amd/csrc/host/file_settings.hpp), the counterpart
of the cv::FileStorage the reference driver reads (eventCameraCalib.cpp:114-207)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_settings_file_shapes(tmp_path):
    exe = str(tmp_path / "test_settings")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_settings.cpp")])
    f = tmp_path / "s.yaml"
    f.write_text('%YAML:1.0\n---\n# comment line\nStartTime: 5\nMotionTimeStep: 5e-4   # trailing comment\nCamera.width: 346\n'
                 'FrameEventNumThreshold:   4000\nuseSO3: 0\nName: "a # b"\nViewer.Facing: [ 1,0,0,0,1,0,0,0,1 ]\n\n')
    out = subprocess.run([exe, str(f)], capture_output=True, text=True, check=True).stdout.split()
    assert out[1] == "0.00050000000000000001" and float(out[3]) == 5.0 and out[5] == "346" and out[7] == "4000"
    assert out[9] == "0" and out[11] == "1" and out[13] == "1"          # useSO3 read; NoSuchKey untouched; EndTime absent
    assert " ".join(out[15:18]) == "[a # b]" and out[19:] == ["9", "1.0", "1.0"]
