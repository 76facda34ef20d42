"""GPU: the C++ shims (eventcalib_amd/csrc/host) compile against include/ecal.h + libecal.so and
behave like the reference classes they replace."""
import os
import subprocess

import pytest

import synth_stream as SS

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# a settings file with the keys the reference driver reads (eventCameraCalib.cpp:114-207, parameters.hpp:15-43,
# CirclesEventFrame.cpp:42-48) and the values of the shipped configuration; PieceNum is the build's
# own key (the reference derives the piece count from the host's core count)
SETTINGS_YAML = """%%YAML:1.0
# event stream
StartTime: %(start)s
EndTime: %(end)s      # optional
MotionTimeStep: 5e-4
FrameEventNumThreshold: 4000
Camera.width: 346
Camera.height: 260
Is_Pattern_Asymmetric: 1
BoardSize_Rows: 9
BoardSize_Cols: 4
Square_Size: 5.5
Circles_Radius: 1.75
Calibrate_NrOfFrameToUse: 200
Calibrate_UseFisheyeModel: 0
Calibrate_FixAspectRatio: 1
Calibrate_AssumeZeroTangentialDistortion: 1
Calibrate_FixPrincipalPointAtTheCenter: 1
Fix_K1: 0
Fix_K2: 0
Fix_K3: 0
Fix_K4: 1
Fix_K5: 1
dbscan_eps: 4
dbscan_startMinSample: 2
clusterMinSample: 5
knn_num: 3
fitCircle: 0
useSO3: 0
reduceMap: 0
Viewer.Facing: [ 1,0,0,0,1,0,0,0,1 ]
Qbs: [ 0, 0, 0, 1 ]
PieceNum: 30
"""


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
    # the pieces are independent: several host threads with their own contexts select the same keyframes
    kf4 = detect_keyframes(DetectPipeline(ctx), buf.cuda(), 5e-4, 4000, 4, float(t[0]), float(t[-1]), n_threads=3)
    assert np.array_equal(kf4["time"], kf["time"]) and np.array_equal(kf4["features"], kf["features"])
    ctx.close()


@pytest.mark.parametrize("fisheye", [0, 1])
def test_cpp_event_calib_ini(tmp_path, fisheye):
    """EventCalibIni::cvCalibration shim: frame selection (step = frames / NumOfFrameToUse), calibration, batched PnP,
    checkPose and the rectify hook, on synthetic keyframes with known camera and poses."""
    import numpy as np
    import synth_calib as SC
    exe = str(tmp_path / "test_calib_shim")
    lib_dir = os.path.join(ROOT, "eventcalib_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_calib_shim.cpp"),
                           "-L" + lib_dir, "-lecal", "-Wl,-rpath," + lib_dir, "-lpthread"])
    V = 50
    obj, img, rv, tv = SC.make_views(V, fisheye, seed=31)
    ts = 5.0 + 1.0 * np.arange(V)             # far apart in time: every pose passes checkPose
    ts[10] = ts[9] + 1e-4                     # ... except one implausibly fast jump
    vf = str(tmp_path / "views.bin")
    np.concatenate([[V, obj.shape[0], fisheye], img.ravel(), ts]).astype(np.float64).tofile(vf)
    out = subprocess.run([exe, vf], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.splitlines()
    head = lines[0].split()
    gt = SC.GT_FISHEYE if fisheye else SC.GT_PINHOLE
    assert head[1] == "1" and float(head[3]) < 1e-4 and int(head[5]) == 20        # float-narrowed pixels: rms ~1e-5
    K = np.array([float(x) for x in lines[1].split()[1:]])
    assert np.abs(K / gt[:4] - 1).max() < 1e-5
    dist = np.array([float(x) for x in lines[2].split()[1:]])
    assert np.abs(dist - (gt[5:9] if fisheye else gt[4:12])).max() < 1e-3
    n_check, n_rect, n_acc = int(head[9]), int(head[11]), int(head[7])
    assert n_check == 1 and n_rect == (V - 1) // 7 and n_acc == V - n_check - n_rect
    p0 = [float(x) for x in lines[3].split()[1:]]
    assert np.abs(np.array(p0[:3]) - tv[0]).max() < 5e-3 and abs(p0[3] - SC.rodrigues(rv[0])[0, 0]) < 1e-4


def test_cpp_driver_chain(tmp_path):
    """The whole reference driver on the C++ shims (keyframes -> cvCalibration + rectify -> EventCalibSpline -> TUM file)
    agrees with the Python mirror (eventcalib_amd/calibrate.py) on the same .bin file and recovers the camera."""
    import numpy as np
    import torch
    import eventcalib_amd
    from eventcalib_amd.calibrate import calibrate_stream
    # the product's executable (eventcalib_amd/csrc/Makefile, target `driver`): built with the library, in-tree
    exe = os.path.join(ROOT, "eventcalib_amd", "unit_test_eventCameraCalib")
    if not os.path.exists(exe):      # (it travels to the GPU box prebuilt, like libecal.so)
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "eventcalib_amd", "csrc"), "driver"])
    SS.TRAJECTORY = "orbit"
    try:
        n = 2_000_000
        buf = SS.make_stream(n, rate=1.0e6, t_start=5.0, device="cpu", seed=21)
    finally:
        SS.TRAJECTORY = "hover"
    binf = str(tmp_path / "events.bin")
    buf.numpy().tofile(binf)
    yamlf = str(tmp_path / "settings.yaml")
    open(yamlf, "w").write(SETTINGS_YAML % dict(start=5, end=8))
    out = subprocess.run([exe, yamlf, binf, str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.splitlines()
    init = lines[1].split()
    ref = [float(v) for v in lines[2].split()[1:10]]
    ctx = eventcalib_amd.Context(0)
    py = calibrate_stream(ctx, buf.cuda(), 5.0, 5.0 + (n - 1) / 1e6)
    ctx.close()
    assert int(lines[0].split()[1]) == py["keyframes"]
    assert abs(float(init[2]) - py["init"]["intr"][0]) < 1e-6 * py["init"]["intr"][0]           # same init calibration
    assert int(init[9]) == py["init"]["accepted"]
    assert np.abs(np.array(ref[:4]) - py["intrinsics"][:4]).max() < 1e-3                         # same refined camera
    assert abs(ref[0] / SS.FX - 1) < 2e-3 and abs(ref[2] - (SS.CX - 0.5)) < 0.3
    traj = np.loadtxt(str(tmp_path / "TrajectoryByEvent.txt"))
    assert traj.shape[1] == 8 and len(traj) == len(py["trajectory"])
    assert np.abs(traj - py["trajectory"]).max() < 1e-3
    # saveDir/image/<std::to_string(time stamp)>.png of every keyframe in the map (eventCameraCalib.cpp:214-227): sensor-sized RGB
    # files a PNG decoder accepts, cluster pixels + green candidate circles + white features on black
    import glob
    import test_png_writer as TP
    pngs = sorted(glob.glob(str(tmp_path / "image" / "*.png")))
    assert len(pngs) == int(init[9]) and lines[3].split() == ["images", str(len(pngs))]
    png_t = np.sort([float(os.path.basename(f)[:-4]) for f in pngs])
    assert all(np.abs(png_t - t).min() < 1e-6 for t in traj[:, 0])        # (std::to_string keeps six decimals)
    img = TP.decode_png(pngs[len(pngs) // 2])
    assert img.shape == (260, 346, 3)
    green = (img == [0, 255, 0]).all(axis=2).sum()
    white = (img == [255, 255, 255]).all(axis=2).sum()
    cluster = ((img[:, :, 0] == 200) | (img[:, :, 0] == 100)).sum()
    # (the features are drawn over the candidates they were taken from: green is left only where a candidate is not in the grid)
    assert white > 36 * 20 and green >= 0 and cluster > 500 and (img.sum(axis=2) == 0).mean() > 0.8, (green, white, cluster)
    # "batch": rectifyFeatures of all keyframes in one device pass (ecal_rectify_keyframes), the file read in one piece —
    # the same chain, the same numbers, plus a time per stage (what bench.py's end_to_end leg runs at 50 M events)
    out2 = subprocess.run([exe, yamlf, binf, str(tmp_path), "batch"], capture_output=True, text=True, timeout=600)
    assert out2.returncode == 0, out2.stdout + out2.stderr
    l2 = [l for l in out2.stdout.splitlines() if not l.startswith("stage ")]
    assert l2[0] == lines[0] and l2[1] == lines[1]                 # keyframes, init calibration + accepted / discarded counts
    ref2 = [float(v) for v in l2[2].split()[1:10]]
    assert np.abs(np.array(ref2) - np.array(ref)).max() < 1e-9 * 400 and l2[2].split()[10:] == lines[2].split()[10:]
    stages = dict(l.split()[1:3] for l in out2.stdout.splitlines() if l.startswith("stage "))
    assert set(stages) == {"runtime_init", "load_file", "upload", "keyframe_search", "init_calibration_pnp_rectify", "spline_fit_association_lm",
                           "save_trajectory"}


def test_cv_find_circles_grid_shim(tmp_path):
    """host/cv_calib.hpp: cv::findCirclesGrid(points, Size(cols, rows), centers, CALIB_CB_ASYMMETRIC_GRID [| CLUSTERING]) in the
    signature of the reference's vendored finder (cv_calib/include/cv_calib.hpp:19-21), called as CirclesEventFrame.cpp:332-353
    calls it: projected circle centres in shuffled order (+ noise, + a spurious point outside the pattern) come back ordered row
    by row — index i * cols + j = model point ((2 j + i % 2) s, i s) —, an incomplete list gives false."""
    import numpy as np
    import torch
    exe = str(tmp_path / "test_cv_calib_shim")
    lib_dir = os.path.join(ROOT, "eventcalib_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_cv_calib_shim.cpp"),
                           "-I" + os.path.join(ROOT, "include"), "-L" + lib_dir, "-lecal", "-Wl,-rpath," + lib_dir, "-lpthread"])
    rng = np.random.default_rng(5)
    lm = SS.landmarks()
    text, want = [], []
    for k, t in enumerate(np.linspace(5.0, 8.0, 12)):
        R, C = SS.pose(torch.tensor([t], dtype=torch.float64))
        gt = SS.project(lm, R.expand(36, 3, 3), C.expand(36, 3)).numpy() + rng.normal(0, 0.7, size=(36, 2))
        pts = gt.copy()
        if k % 3 == 1:
            pts = np.vstack([pts, [[8.0, 8.0]]])                 # a spurious candidate in the image corner
        if k % 3 == 2:
            pts = np.delete(pts, 17, axis=0)                      # a circle missing: no grid
        perm = rng.permutation(len(pts))
        pts = pts[perm]
        text.append("%d\n%s\n" % (len(pts), "\n".join("%.6f %.6f" % (x, y) for x, y in pts)))
        want.append(None if k % 3 == 2 else [int(np.flatnonzero(perm == m)[0]) for m in range(36)])
    out = subprocess.run([exe], input="".join(text), capture_output=True, text=True, check=True).stdout.strip().split("\n")
    assert len(out) == len(want)
    for line, w in zip(out, want):
        if w is None:
            assert line == "none"
        else:
            assert line.split()[0] == "found" and [int(v) for v in line.split()[1:]] == w
