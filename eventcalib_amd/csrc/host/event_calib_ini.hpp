// Counterpart of the reference's EventCalibIni initialisation stage
//   CalibrationSetting / validate()         event_camera_calib/include/opengv2/event_camera_calib/parameters.hpp:29-84
//   EventCalibIni::calcBoardCornerPositions event_camera_calib/src/EventCalibIni.cpp:99-115
//   EventCalibIni::cvCalibration            event_camera_calib/src/EventCalibIni.cpp:149-325
//   EventCalibIni::checkPose                event_camera_calib/src/EventCalibIni.cpp:327-347
// on top of libecal.so.  The OpenCV calls of cvCalibration (calibrateCamera / fisheye::calibrate, solvePnPRansac,
// Rodrigues) become ecal_calibrate_views and ONE ecal_pnp_batch over all keyframes; the map / viewer object graph
// of the reference is replaced by plain vectors (out of scope).
#ifndef ECAL_HOST_EVENT_CALIB_INI_HPP_
#define ECAL_HOST_EVENT_CALIB_INI_HPP_

#include <array>
#include <functional>

#include "multi_process.hpp"

namespace opengv2 {

struct CalibrationSetting {
    typedef std::shared_ptr<CalibrationSetting> Ptr;
    CirclePatternParameters::Ptr circlePatternParameters = std::make_shared<CirclePatternParameters>();
    int NumOfFrameToUse = 200;            // Calibrate_NrOfFrameToUse
    float aspectRatio = 1;                // Calibrate_FixAspectRatio
    bool calibZeroTangentDist = true;     // Calibrate_AssumeZeroTangentialDistortion
    bool calibFixPrincipalPoint = true;   // Calibrate_FixPrincipalPointAtTheCenter
    bool useFisheye = false;              // Calibrate_UseFisheyeModel
    bool fixK1 = false, fixK2 = false, fixK3 = false, fixK4 = true, fixK5 = true;
    uint32_t flag = 0;                    // ECAL_CALIB_* (the reference keeps cv::CALIB_* bits here)

    CalibrationSetting() = default;
    explicit CalibrationSetting(const FileSettings &node) {   // parameters.hpp:32-46
        circlePatternParameters = std::make_shared<CirclePatternParameters>(node);
        node["Calibrate_FixAspectRatio"] >> aspectRatio;
        node["Calibrate_AssumeZeroTangentialDistortion"] >> calibZeroTangentDist;
        node["Calibrate_FixPrincipalPointAtTheCenter"] >> calibFixPrincipalPoint;
        node["Calibrate_UseFisheyeModel"] >> useFisheye;
        node["Fix_K1"] >> fixK1;
        node["Fix_K2"] >> fixK2;
        node["Fix_K3"] >> fixK3;
        node["Fix_K4"] >> fixK4;
        node["Fix_K5"] >> fixK5;
        node["Calibrate_NrOfFrameToUse"] >> NumOfFrameToUse;
        validate();
    }

    void validate() {  // parameters.hpp:47-69
        flag = 0;
        if (calibFixPrincipalPoint) flag |= ECAL_CALIB_FIX_PRINCIPAL_POINT;
        if (calibZeroTangentDist) flag |= ECAL_CALIB_ZERO_TANGENT_DIST;
        if (aspectRatio) flag |= ECAL_CALIB_FIX_ASPECT_RATIO;
        if (fixK1) flag |= ECAL_CALIB_FIX_K1;
        if (fixK2) flag |= ECAL_CALIB_FIX_K2;
        if (fixK3) flag |= ECAL_CALIB_FIX_K3;
        if (fixK4) flag |= ECAL_CALIB_FIX_K4;
        if (fixK5) flag |= ECAL_CALIB_FIX_K5;
        flag |= ECAL_CALIB_FIX_K6;
        if (useFisheye) {  // the fisheye model has its own enum, so overwrite the flags
            flag = ECAL_CALIB_FIX_SKEW | ECAL_CALIB_RECOMPUTE_EXTRINSIC;
            if (fixK1) flag |= ECAL_CALIB_FIX_K1;
            if (fixK2) flag |= ECAL_CALIB_FIX_K2;
            if (fixK3) flag |= ECAL_CALIB_FIX_K3;
            if (fixK4) flag |= ECAL_CALIB_FIX_K4;
            if (calibFixPrincipalPoint) flag |= ECAL_CALIB_FIX_PRINCIPAL_POINT;
        }
    }
};

class EventCalibIni {
public:
    struct FramePose {       // what the reference stores with bf->setPose(twb, unitQwb) plus the PnP output
        double Rsw[9];       // camera <- board, row-major (cv::Rodrigues(rvec))
        double tsw[3];
        double twb[3];       // = Rsw^T (tsb - tsw) with the default body->sensor extrinsics (identity)
        std::vector<int> outlierIdxs;
    };
    struct Result {
        bool ok = false;
        double K[4] = {0, 0, 0, 0};         // fx fy cx cy
        std::vector<double> distCoeffs;     // 8 (k1 k2 p1 p2 k3 k4 k5 k6) or 4 (fisheye k1..k4)
        double rms = 0;
        int fisheyeStart = -1;   // Calibrate_UseFisheyeModel: 0 the reference's own start, 1 the radial guess was needed (ecal_calibrate_fisheye_views)
        std::vector<size_t> usedFrames;     // keyframes that entered the calibration
        std::vector<size_t> acceptedFrames; // keyframes that passed checkPose and rectifyFeatures, time order
        std::vector<FramePose> poses;       // one per keyframe (valid for every keyframe the PnP solved)
        int discardedByCheckPose = 0, discardedByRectify = 0;
        double intr[12];
    };
    // optional hook = cf->rectifyFeatures(outlierIdxs, Rsw, tsw) of the frame behind keyframe i (EventCalibIni.cpp:294)
    typedef std::function<bool(size_t, const FramePose &)> RectifyFn;
    // the same for ALL keyframes at once, called once after the PnP (rectifyFeatures of a frame depends on its own pose only,
    // not on which frames were accepted before it): ok[f] = its return value for keyframe f (ecal_rectify_keyframes)
    typedef std::function<void(const std::vector<FramePose> &, const std::vector<char> &pnp_ok, std::vector<char> &ok)> BatchRectifyFn;

    EventCalibIni(CalibrationSetting::Ptr setting, double motionTimeStep)
        : calibrationSetting_(std::move(setting)), motionTimeStep_(motionTimeStep) {}

    void calcBoardCornerPositions(std::vector<std::array<double, 3>> &corners) const {
        corners.clear();
        const auto &p = *calibrationSetting_->circlePatternParameters;
        for (int i = 0; i < p.rows; i++)
            for (int j = 0; j < p.cols; j++)
                corners.push_back({(double) (float) ((p.isAsymmetric ? (2 * j + i % 2) : j) * p.squareSize),
                                   (double) (float) (i * p.squareSize), 0.0});  // cv::Point3f
    }

    // EventCalibIni::checkPose: translational / angular speed between consecutive accepted keyframes
    bool checkPose(const FramePose &ref, double tRef, const FramePose &cur, double tCur) const {
        const double duration = tCur - tRef;
        double d[3] = {cur.twb[0] - ref.twb[0], cur.twb[1] - ref.twb[1], cur.twb[2] - ref.twb[2]};
        // Qwb^-1 = Rsw (Rwb = Rsw^T): rotating by it keeps the norm, so v_t is just |d| / duration
        const double v_t = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) / duration;
        // angular distance between the two orientations: angle of Rsw_cur Rsw_ref^T
        double tr = 0;
        for (int i = 0; i < 3; i++)
            for (int k = 0; k < 3; k++) tr += cur.Rsw[3 * i + k] * ref.Rsw[3 * i + k];
        const double v_R = std::fabs(std::acos(std::max(-1.0, std::min(1.0, (tr - 1) * 0.5))) / duration);
        return v_t < (2.5e-1 / motionTimeStep_) * 2 && v_R < (5e-4 * M_PI) * 2 / motionTimeStep_;
    }

    bool cvCalibration(const std::vector<KeyFrame> &keyframes, double width, double height, Result &res,
                       const RectifyFn &rectify = RectifyFn(), const BatchRectifyFn &batchRectify = BatchRectifyFn()) {
        res = Result();
        auto &cs = *calibrationSetting_;
        const int frameNum = (int) keyframes.size();
        if (frameNum == 0) return false;
        int use = cs.NumOfFrameToUse, step = frameNum / std::max(use, 1);
        if (step == 0) {  // :163-167
            use = frameNum;
            step = 1;
        }
        std::vector<std::array<double, 3>> corners;
        calcBoardCornerPositions(corners);
        const uint32_t n = (uint32_t) corners.size();
        std::vector<double> obj(3 * (size_t) n), img;
        for (uint32_t i = 0; i < n; i++)
            for (int k = 0; k < 3; k++) obj[3 * i + k] = corners[i][k];
        for (int c = 0, idx = 0; c < use; c++, idx += step) {  // :170-181; narrowed to float as cv::Point2f
            res.usedFrames.push_back((size_t) idx);
            for (const auto &f : keyframes[idx].features) {
                img.push_back((double) (float) f.location[0]);
                img.push_back((double) (float) f.location[1]);
            }
        }
        ecal_ctx *ctx = ecal_host::thread_ctx();
        ecal_calib_options opt;
        ecal_calib_default_options(&opt);
        opt.model = cs.useFisheye ? 1 : 0;
        opt.flags = cs.flag;
        opt.aspect_ratio = cs.aspectRatio;
        ecal_calib_result cr;
        int rc;
        if (cs.useFisheye) {
            // the library's one start procedure (ecal_calibrate_fisheye_views): the reference's own start first, the radial
            // model's focal lengths as a guess only when that fails; res.fisheyeStart says which
            int start_used = 0;
            rc = ecal_calibrate_fisheye_views(ctx, obj.data(), n, img.data(), (uint32_t) use, width, height, &opt, &cr, nullptr, nullptr,
                                              nullptr, &start_used);
            res.fisheyeStart = start_used;
        } else {
            rc = ecal_calibrate_views(ctx, obj.data(), n, img.data(), (uint32_t) use, width, height, &opt, &cr, nullptr, nullptr, nullptr);
        }
        if (rc == ECAL_ERR_INVALID) return false;  // degenerate views: what cv::calibrateCamera reports by throwing
        if (rc != ECAL_OK) throw std::runtime_error(std::string("ecal_calibrate_views: ") + ecal_last_error(ctx));
        bool ok = std::isfinite(cr.rms);
        for (int j = 0; j < 12; j++) ok = ok && std::isfinite(cr.intr[j]);  // cv::checkRange
        res.ok = ok;
        res.rms = cr.rms;
        for (int j = 0; j < 12; j++) res.intr[j] = cr.intr[j];
        for (int j = 0; j < 4; j++) res.K[j] = cr.intr[j];
        if (cs.useFisheye) res.distCoeffs.assign(cr.intr + 5, cr.intr + 9);
        else res.distCoeffs.assign(cr.intr + 4, cr.intr + 12);
        if (!ok) return false;
        // solvePnPRansac for every keyframe (:252-259), one batched call
        std::vector<double> all((size_t) frameNum * n * 2), pose((size_t) frameNum * 6);
        std::vector<uint32_t> inl((size_t) frameNum * n), okf(frameNum);
        for (int f = 0; f < frameNum; f++)
            for (uint32_t i = 0; i < n; i++) {
                all[((size_t) f * n + i) * 2] = (double) (float) keyframes[f].features[i].location[0];
                all[((size_t) f * n + i) * 2 + 1] = (double) (float) keyframes[f].features[i].location[1];
            }
        rc = ecal_pnp_batch(ctx, obj.data(), n, all.data(), nullptr, (uint32_t) frameNum, opt.model, cr.intr, 4.0, 3, 0, pose.data(),
                            inl.data(), nullptr, okf.data());
        if (rc != ECAL_OK) throw std::runtime_error(std::string("ecal_pnp_batch: ") + ecal_last_error(ctx));
        res.poses.resize(frameNum);
        for (int f = 0; f < frameNum; f++) {
            FramePose &fp = res.poses[f];
            rodrigues(&pose[6 * (size_t) f], fp.Rsw);
            for (int k = 0; k < 3; k++) fp.tsw[k] = pose[6 * (size_t) f + 3 + k];
            for (int k = 0; k < 3; k++) fp.twb[k] = -(fp.Rsw[k] * fp.tsw[0] + fp.Rsw[3 + k] * fp.tsw[1] + fp.Rsw[6 + k] * fp.tsw[2]);
            for (uint32_t i = 0; i < n; i++)
                if (!inl[(size_t) f * n + i]) fp.outlierIdxs.push_back((int) i);
        }
        std::vector<char> batch_ok;
        if (batchRectify) {
            std::vector<char> pnp_ok(frameNum);
            for (int f = 0; f < frameNum; f++) pnp_ok[f] = okf[f] ? 1 : 0;
            batch_ok.assign(frameNum, 0);
            batchRectify(res.poses, pnp_ok, batch_ok);
        }
        long last = -1;
        for (int f = 0; f < frameNum; f++) {
            FramePose &fp = res.poses[f];
            if (!okf[f]) {
                res.discardedByCheckPose++;
                continue;
            }
            if (last >= 0 && !checkPose(res.poses[last], keyframes[last].timeStamp, fp, keyframes[f].timeStamp)) {
                res.discardedByCheckPose++;
                continue;
            }
            if (batchRectify ? !batch_ok[f] : (rectify && !rectify((size_t) f, fp))) {
                res.discardedByRectify++;
                continue;
            }
            res.acceptedFrames.push_back((size_t) f);
            last = f;
        }
        return true;
    }

    static void rodrigues(const double *v, double *R) {  // cv::Rodrigues(rvec, R)
        const double th = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        if (th < 2.220446049250313e-16) {
            for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
            return;
        }
        const double x = v[0] / th, y = v[1] / th, z = v[2] / th, c = std::cos(th), s = std::sin(th), c1 = 1 - c;
        R[0] = c + c1 * x * x; R[1] = c1 * x * y - s * z; R[2] = c1 * x * z + s * y;
        R[3] = c1 * x * y + s * z; R[4] = c + c1 * y * y; R[5] = c1 * y * z - s * x;
        R[6] = c1 * x * z - s * y; R[7] = c1 * y * z + s * x; R[8] = c + c1 * z * z;
    }

private:
    CalibrationSetting::Ptr calibrationSetting_;
    double motionTimeStep_;
};

}  // namespace opengv2

#endif  // ECAL_HOST_EVENT_CALIB_INI_HPP_
