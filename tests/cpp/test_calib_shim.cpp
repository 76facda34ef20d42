// Exercises host/event_calib_ini.hpp the way the reference driver uses EventCalibIni::cvCalibration
// (event_camera_calib/test/eventCameraCalib.cpp:196-200).  Built and run by tests/test_gpu_shims.py.
//   usage: test_calib_shim views.bin   (f64: V, n, fisheye, then V x n x 2 pixel coordinates, then V timestamps)
#include <cstdio>
#include <fstream>

#include "../../eventcalib_amd/csrc/host/event_calib_ini.hpp"

int main(int argc, char **argv) {
    using namespace opengv2;
    if (argc < 2) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    double hdr[3];
    f.read(reinterpret_cast<char *>(hdr), sizeof(hdr));
    const int V = (int) hdr[0], n = (int) hdr[1];
    std::vector<double> img((size_t) V * n * 2), ts(V);
    f.read(reinterpret_cast<char *>(img.data()), img.size() * sizeof(double));
    f.read(reinterpret_cast<char *>(ts.data()), ts.size() * sizeof(double));
    if (!f) return 3;
    std::vector<KeyFrame> kfs(V);
    for (int v = 0; v < V; v++) {
        kfs[v].timeStamp = ts[v];
        kfs[v].duration = {ts[v] - 7.5e-4, ts[v] + 7.5e-4};
        kfs[v].eventsNum = 1500;
        for (int i = 0; i < n; i++)
            kfs[v].features.push_back(CirclesEventFrame::CalibCircle{Vector2d{{img[((size_t) v * n + i) * 2], img[((size_t) v * n + i) * 2 + 1]}}, 6.0});
    }
    auto cs = std::make_shared<CalibrationSetting>();   // the shipped example.yaml
    cs->useFisheye = hdr[2] != 0;
    cs->NumOfFrameToUse = 20;
    if (cs->useFisheye) cs->fixK4 = true;
    cs->validate();
    EventCalibIni ini(cs, 5e-4);
    EventCalibIni::Result res;
    size_t rectified = 0;
    const bool ok = ini.cvCalibration(kfs, 346, 260, res, [&](size_t, const EventCalibIni::FramePose &) { return ++rectified % 7 != 0; });
    std::printf("ok %d rms %.12g used %zu accepted %zu checkPose %d rectify %d\n", (int) ok, res.rms, res.usedFrames.size(),
                res.acceptedFrames.size(), res.discardedByCheckPose, res.discardedByRectify);
    std::printf("K %.12g %.12g %.12g %.12g\n", res.K[0], res.K[1], res.K[2], res.K[3]);
    std::printf("dist");
    for (double d : res.distCoeffs) std::printf(" %.12g", d);
    std::printf("\n");
    for (size_t k = 0; k < 3 && k < res.poses.size(); k++)
        std::printf("pose %.12g %.12g %.12g %.12g\n", res.poses[k].tsw[0], res.poses[k].tsw[1], res.poses[k].tsw[2], res.poses[k].Rsw[0]);
    return ok ? 0 : 1;
}
