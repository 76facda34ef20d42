// unit_test_eventCameraCalib: the reference driver's main (event_camera_calib/test/eventCameraCalib.cpp:99-233) on the C++ shims:
// stream -> container -> keyframe search -> EventCalibIni::cvCalibration (+ rectifyFeatures per keyframe) ->
// EventCalibSpline -> TrajectoryByEvent.txt.  Built by eventcalib_amd/csrc/Makefile (`make driver`, part of `all`) into
// eventcalib_amd/unit_test_eventCameraCalib next to libecal.so; run by tests/test_gpu_shims.py and bench.py's end_to_end leg.
//   usage: unit_test_eventCameraCalib settings.yaml events.bin saveDir [batch]     (the reference's argv, eventCameraCalib.cpp:105-110)
// "batch": rectifyFeatures of all keyframes in one device pass (ecal_rectify_keyframes) instead of one CirclesEventFrame per
// keyframe, the file read in one piece, and a "stage <name> <seconds>" line per stage of the chain (bench.py's end_to_end leg).
#include <chrono>
#include <cstdio>

#include <filesystem>
#include <thread>
#include <algorithm>

#include "event_calib_spline.hpp"
#include "png_writer.hpp"

int main(int argc, char **argv) {
    using namespace opengv2;
    const bool batch = argc == 5 && std::string(argv[4]) == "batch";
    double t_mark = 0;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto stage = [&](const char *name) {
        const double t = now();
        if (batch) std::printf("stage %s %.6f\n", name, t - t_mark);
        t_mark = t;
    };
    t_mark = now();
    if (argc != 4 && !batch) {
        std::fprintf(stderr, "Usage: unit_test_eventCameraCalib settingFilePath binFilePath SavePath\n");
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
    if (batch) {   // the same loop as one call: file -> HBM, reads and upload overlapped (EventContainer::loadFile)
        (void) ecal_host::thread_ctx();   // HIP runtime + context start-up (0.2 - 0.3 s in a cold process): a stage of its own,
        stage("runtime_init");            // not part of reading the file
        es.close();
        container->loadFile(argv[2], startTime, customEnd, endTimeSetting);
    }
    while (!es.isEnd()) {                                    // :154-163
        if (customEnd && es.current().timeStamp() >= endTimeSetting) break;
        if (es.current().timeStamp() >= startTime) container->emplace(es.current());
        es.next();
    }
    es.close();
    stage("load_file");
    (void) container->device();                              // the one upload (the reference fills its multimap here)
    stage("upload");
    const double endTime = container->lastTime();            // :165
    const int frameEventNumThreshold = fsSettings["FrameEventNumThreshold"];   // :168
    auto pattern = cs->circlePatternParameters;
    CirclesEventFrame::Params fp(fsSettings);                // :171
    // the reference's piece count (:172-173): 5 * (std::thread::hardware_concurrency() - 2) — the keyframe set depends on it (the
    // pieces' boundaries), so the same binary on the same box gives what the reference would; PieceNum (a key of this build,
    // optional) pins it where a result must not depend on the host (the tests do)
    int pieceNum = 5 * std::max(1, (int) std::thread::hardware_concurrency() - 2);
    fsSettings["PieceNum"] >> pieceNum;
    // GateMode (a key of this build, optional): 1 = ECAL_GATE_SHARED_MAP, the reference's semantics with one worker thread (the
    // default: one keyframe map for all pieces, TrackingBase.cpp:16-46, EventCalibIni.cpp:26-36); 0 = ECAL_GATE_OWN_PIECE, the
    // faster schedule-free gate (every piece's first success ungated)
    int gateMode = ECAL_GATE_SHARED_MAP;
    fsSettings["GateMode"] >> gateMode;
    std::vector<KeyFrame> kfs = detect_keyframes_device(*container, pattern, fp, step, frameEventNumThreshold, pieceNum, startTime, endTime, gateMode);
    std::printf("keyframes %zu\n", kfs.size());
    stage("keyframe_search");
    EventCalibIni ini(cs, step);
    EventCalibIni::Result res;
    std::vector<EventCalibSpline::Frame> frames;
    const size_t n_circ = (size_t) (pattern->rows * pattern->cols);
    // the rectify hook runs after K and the distortion are known (res is filled before the PnP loop)
    auto rectify = [&](size_t f, const EventCalibIni::FramePose &p) {
        CirclesEventFrame cf(container, kfs[f].duration, pattern, fp);
        if (!cf.extractFeatures()) return false;
        CirclesEventFrame::Camera cam{res.K[0], res.K[1], res.K[2], res.K[3], {0, 0, 0, 0, 0}};
        for (size_t i = 0; i < 5 && i < res.distCoeffs.size(); i++) cam.distCoeffs[i] = res.distCoeffs[i];
        cam.fisheye = cs->useFisheye;   // (4 coefficients k1..k4 then)
        double R[9], t[3];
        for (int i = 0; i < 9; i++) R[i] = p.Rsw[i];
        for (int i = 0; i < 3; i++) t[i] = p.tsw[i];
        if (!cf.rectifyFeatures({}, R, t, cam)) return false;
        frames.push_back(EventCalibSpline::makeFrame(kfs[f].timeStamp, p, cf.features(), cf.featureLandmark(), n_circ));
        return true;
    };
    // ... or for all keyframes at once, on the device
    std::vector<double> rect_feat;
    std::vector<uint32_t> rect_valid;
    auto rectify_all = [&](const std::vector<EventCalibIni::FramePose> &poses, const std::vector<char> &pnp_ok, std::vector<char> &ok) {
        const uint32_t F = (uint32_t) kfs.size();
        std::vector<double> dur(2 * (size_t) F), pose(12 * (size_t) F), lm(3 * n_circ);
        for (uint32_t f = 0; f < F; f++) {
            dur[2 * f] = kfs[f].duration.first;
            dur[2 * f + 1] = kfs[f].duration.second;
            for (int i = 0; i < 9; i++) pose[12 * (size_t) f + i] = poses[f].Rsw[i];
            for (int i = 0; i < 3; i++) pose[12 * (size_t) f + 9 + i] = poses[f].tsw[i];
        }
        for (int i = 0; i < pattern->rows; i++)
            for (int j = 0; j < pattern->cols; j++) {
                double *o = &lm[3 * (size_t) (i * pattern->cols + j)];
                o[0] = (float) ((pattern->isAsymmetric ? (2 * j + i % 2) : j) * pattern->squareSize);
                o[1] = (float) (i * pattern->squareSize);
                o[2] = 0;
            }
        CirclesEventFrame probe(container, kfs[0].duration, pattern, fp);
        ecal_detect_params dp;
        dp.dbscan_eps = fp.dbscan_eps;
        dp.dbscan_min_samples = (uint32_t) fp.dbscan_startMinSample;
        dp.cluster_min_sample = (uint32_t) fp.clusterMinSample;
        dp.need_clusters = (uint32_t) n_circ;
        dp.circle_radius_threshold = probe.circleRadiusThreshold();
        dp.fit_circle = fp.fitCircle ? 1 : 0;
        dp.knn_num = (uint32_t) fp.knn_num;
        dp.rows = (uint32_t) pattern->rows;
        dp.cols = (uint32_t) pattern->cols;
        ecal_rectify_params rp;
        rp.fx = res.K[0], rp.fy = res.K[1], rp.cx = res.K[2], rp.cy = res.K[3];
        for (size_t i = 0; i < 5; i++) rp.dist[i] = i < res.distCoeffs.size() ? res.distCoeffs[i] : 0.0;
        rp.width = container->cameraSize[0], rp.height = container->cameraSize[1];
        rp.rows = (uint32_t) pattern->rows, rp.cols = (uint32_t) pattern->cols;
        rp.asymmetric = pattern->isAsymmetric ? 1 : 0;
        rp.circle_radius = pattern->circleRadius;
        rp.fit_circle = fp.fitCircle ? 1 : 0;
        rp.model = cs->useFisheye ? 1 : 0;
        rect_feat.resize(3 * n_circ * (size_t) F);
        rect_valid.resize(n_circ * (size_t) F);
        std::vector<uint32_t> info(2 * (size_t) F);
        const int rc = ecal_rectify_keyframes(ecal_host::thread_ctx(), container->device(), dur.data(), F, &dp, pose.data(), lm.data(), &rp,
                                              rect_feat.data(), rect_valid.data(), info.data());
        if (rc != ECAL_OK) throw std::runtime_error(std::string("ecal_rectify_keyframes: ") + ecal_last_error(ecal_host::thread_ctx()));
        for (uint32_t f = 0; f < F; f++) ok[f] = pnp_ok[f] && info[2 * f] != 0;
    };
    if (!ini.cvCalibration(kfs, container->cameraSize[0], container->cameraSize[1], res, batch ? EventCalibIni::RectifyFn() : rectify,
                           batch ? EventCalibIni::BatchRectifyFn(rectify_all) : EventCalibIni::BatchRectifyFn()))
        return 1;
    if (batch)   // the accepted keyframes' frames from the batch's arrays
        for (size_t f : res.acceptedFrames) {
            std::vector<CirclesEventFrame::CalibCircle> feats;
            std::vector<int> lmk;
            for (size_t k = 0; k < n_circ; k++)
                if (rect_valid[f * n_circ + k]) {
                    const double *p = &rect_feat[3 * (f * n_circ + k)];
                    feats.push_back(CirclesEventFrame::CalibCircle{Vector2d{{p[0], p[1]}}, p[2]});
                    lmk.push_back((int) k);
                }
            frames.push_back(EventCalibSpline::makeFrame(kfs[f].timeStamp, res.poses[f], feats, lmk, n_circ));
        }
    stage("init_calibration_pnp_rectify");
    std::printf("init K %.9g %.9g %.9g %.9g rms %.6g accepted %zu checkPose %d rectify %d\n", res.K[0], res.K[1], res.K[2], res.K[3],
                res.rms, res.acceptedFrames.size(), res.discardedByCheckPose, res.discardedByRectify);
    double dist5[5] = {0, 0, 0, 0, 0};
    for (size_t i = 0; i < 5 && i < res.distCoeffs.size(); i++) dist5[i] = res.distCoeffs[i];
    bool useSO3 = false;
    fsSettings["useSO3"] >> useSO3;                          // :204-206 (reduceMap: experimental in the reference, not restated)
    EventCalibSpline spline(frames, container, pattern, useSO3, step, res.K, dist5, 50, cs->useFisheye);
    const double *x = spline.intrinsics();
    std::printf("refined %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g residuals %zu iterations %d splines %zu\n", x[0], x[1], x[2], x[3],
                x[4], x[5], x[6], x[7], x[8], spline.summary().residuals, spline.summary().iterations, spline.splineNum());
    stage("spline_fit_association_lm");
    spline.saveKeyFrameTrajectoryTUM(std::string(argv[3]) + "/TrajectoryByEvent.txt");   // :212
    stage("save_trajectory");
    if (!batch) {   // :214-227: saveDir/image/<std::to_string(time stamp)>.png of every keyframe in the map
        const std::string dir = std::string(argv[3]) + "/image";
        std::error_code ec;
        std::filesystem::remove_all(dir, ec);
        std::filesystem::create_directories(dir, ec);
        size_t written = 0;
        for (size_t f : res.acceptedFrames) {
            CirclesEventFrame cf(container, kfs[f].duration, pattern, fp);
            (void) cf.extractFeatures();
            if (ecal_host::write_png_rgb(dir + "/" + std::to_string(kfs[f].timeStamp) + ".png", (int) container->cameraSize[0],
                                         (int) container->cameraSize[1], cf.image()))
                written++;
        }
        std::printf("images %zu\n", written);
    }
    return 0;
}
