// The reference driver's main (event_camera_calib/test/eventCameraCalib.cpp:99-233) on the C++ shims:
// stream -> container -> keyframe search -> EventCalibIni::cvCalibration (+ rectifyFeatures per keyframe) ->
// EventCalibSpline -> TrajectoryByEvent.txt.  Built and run by tests/test_gpu_shims.py.
//   usage: test_calib_chain settings.yaml events.bin saveDir      (the reference's argv, eventCameraCalib.cpp:105-110)
#include <cstdio>

#include "../../eventcalib_amd/csrc/host/event_calib_spline.hpp"

int main(int argc, char **argv) {
    using namespace opengv2;
    if (argc != 4) {
        std::fprintf(stderr, "Usage: test_calib_chain settingFilePath binFilePath SavePath\n");
        return 1;
    }
    FileSettings fsSettings(argv[1]);                        // :114-118
    if (!fsSettings.isOpened()) {
        std::fprintf(stderr, "Failed to open settings file at: %s\n", argv[1]);
        return 2;
    }
    auto cs = std::make_shared<CalibrationSetting>(fsSettings);   // :121
    const double step = fsSettings["MotionTimeStep"];        // :126
    const bool customEnd = !fsSettings["EndTime"].isNone();  // :150-153
    const double startTime = fsSettings["StartTime"];
    const double endTimeSetting = customEnd ? (double) fsSettings["EndTime"] : 0.0;
    EventStream es(argv[2]);
    auto container = std::make_shared<EventContainer>();
    int width = fsSettings["Camera.width"], height = fsSettings["Camera.height"];   // :141-142
    container->cameraSize[0] = width;
    container->cameraSize[1] = height;
    while (!es.isEnd()) {                                    // :154-163
        if (customEnd && es.current().timeStamp() >= endTimeSetting) break;
        if (es.current().timeStamp() >= startTime) container->emplace(es.current());
        es.next();
    }
    es.close();
    const double endTime = container->lastTime();            // :165
    const int frameEventNumThreshold = fsSettings["FrameEventNumThreshold"];   // :168
    auto pattern = cs->circlePatternParameters;
    CirclesEventFrame::Params fp(fsSettings);                // :171
    int pieceNum = 30;                                       // (the reference: 5 * (hardware threads - 2), :172-173; the result
    fsSettings["PieceNum"] >> pieceNum;                      //  of the build's own-piece gate depends on it, so the test pins it)
    std::vector<KeyFrame> kfs = detect_keyframes_device(*container, pattern, fp, step, frameEventNumThreshold, pieceNum, startTime, endTime);
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
    bool useSO3 = false;
    fsSettings["useSO3"] >> useSO3;                          // :204-206 (reduceMap: experimental in the reference, not restated)
    EventCalibSpline spline(frames, container, pattern, useSO3, step, res.K, dist5);
    const double *x = spline.intrinsics();
    std::printf("refined %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g residuals %zu iterations %d splines %zu\n", x[0], x[1], x[2], x[3],
                x[4], x[5], x[6], x[7], x[8], spline.summary().residuals, spline.summary().iterations, spline.splineNum());
    spline.saveKeyFrameTrajectoryTUM(std::string(argv[3]) + "/TrajectoryByEvent.txt");   // :212
    return 0;
}
