// The reference driver's main (event_camera_calib/test/eventCameraCalib.cpp:99-233) on the C++ shims:
// stream -> container -> keyframe search -> EventCalibIni::cvCalibration (+ rectifyFeatures per keyframe) ->
// EventCalibSpline -> TrajectoryByEvent.txt.  Built and run by tests/test_gpu_shims.py.
//   usage: test_calib_chain events.bin saveDir
#include <cstdio>

#include "../../eventcalib_amd/csrc/host/event_calib_spline.hpp"

int main(int argc, char **argv) {
    using namespace opengv2;
    if (argc < 3) return 2;
    EventStream es(argv[1]);
    auto container = std::make_shared<EventContainer>();
    while (!es.isEnd()) {
        container->emplace(es.current());
        es.next();
    }
    es.close();
    const double step = 5e-4;
    auto cs = std::make_shared<CalibrationSetting>();  // example.yaml
    cs->validate();
    auto pattern = cs->circlePatternParameters;
    CirclesEventFrame::Params fp;
    std::vector<KeyFrame> kfs = detect_keyframes_device(*container, pattern, fp, step, 4000, 30, container->firstTime(), container->lastTime());
    std::printf("keyframes %zu\n", kfs.size());
    EventCalibIni ini(cs, step);
    EventCalibIni::Result res;
    std::vector<EventCalibSpline::Frame> frames;
    const size_t n_circ = (size_t) (pattern->rows * pattern->cols);
    // the rectify hook runs after K and the distortion are known (res is filled before the PnP loop)
    auto rectify = [&](size_t f, const EventCalibIni::FramePose &p) {
        CirclesEventFrame cf(container, kfs[f].duration, pattern, fp);
        if (!cf.extractFeatures()) return false;
        const CirclesEventFrame::Camera cam{res.K[0], res.K[1], res.K[2], res.K[3],
                                            {res.distCoeffs[0], res.distCoeffs[1], res.distCoeffs[2], res.distCoeffs[3], res.distCoeffs[4]}};
        double R[9], t[3];
        for (int i = 0; i < 9; i++) R[i] = p.Rsw[i];
        for (int i = 0; i < 3; i++) t[i] = p.tsw[i];
        if (!cf.rectifyFeatures({}, R, t, cam)) return false;
        frames.push_back(EventCalibSpline::makeFrame(kfs[f].timeStamp, p, cf.features(), cf.featureLandmark(), n_circ));
        return true;
    };
    if (!ini.cvCalibration(kfs, container->cameraSize[0], container->cameraSize[1], res, rectify)) return 1;
    std::printf("init K %.9g %.9g %.9g %.9g rms %.6g accepted %zu checkPose %d rectify %d\n", res.K[0], res.K[1], res.K[2], res.K[3],
                res.rms, res.acceptedFrames.size(), res.discardedByCheckPose, res.discardedByRectify);
    const double dist5[5] = {res.distCoeffs[0], res.distCoeffs[1], res.distCoeffs[2], res.distCoeffs[3], res.distCoeffs[4]};
    EventCalibSpline spline(frames, container, pattern, false, step, res.K, dist5);
    const double *x = spline.intrinsics();
    std::printf("refined %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g residuals %zu iterations %d splines %zu\n", x[0], x[1], x[2], x[3],
                x[4], x[5], x[6], x[7], x[8], spline.summary().residuals, spline.summary().iterations, spline.splineNum());
    spline.saveKeyFrameTrajectoryTUM(std::string(argv[2]) + "/TrajectoryByEvent.txt");
    return 0;
}
