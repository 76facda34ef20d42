// Counterpart of the reference's EventCalibSpline (continuous-time refinement stage)
//   constructor: gap segmentation, spline initialisation, intrinsics   event_camera_calib/src/EventCalibSpline.cpp:14-113
//   reduceMap (segmentation part)                                      :318-345
//   optimize: event association + Ceres problem                        :115-251
//   updateMap                                                          :253-317
//   SystemBase::saveKeyFrameTrajectoryTUM                              core/system/src/SystemBase.cpp:122-150
// on top of libecal.so: ecal_spline_fit (initial control points), ecal_associate_dev (per-event association on the
// GPU), ecal_solver_* (the Levenberg-Marquardt solve), ecal_spline_eval (poses back at the keyframe times).  The
// map / body-frame object graph of the reference is a plain vector of frames here.
#ifndef ECAL_HOST_EVENT_CALIB_SPLINE_HPP_
#define ECAL_HOST_EVENT_CALIB_SPLINE_HPP_

#include <fstream>
#include <iomanip>

#include "event_calib_ini.hpp"

namespace opengv2 {

class EventCalibSpline {
public:
    struct Frame {
        double timeStamp;
        double twb[3];
        double Qwb[4];                    // x y z w (Eigen coeffs order)
        std::vector<double> circles;      // [n][3] centre x, y, radius in pixels, grid order; NaN rows = erased features
    };
    struct Summary {
        size_t residuals = 0;
        int iterations = 0;
        double initial_cost = 0, final_cost = 0;
    };

    // rotation matrix (row-major) -> quaternion xyzw, Eigen::Quaterniond(Matrix3d)'s branch structure
    static void quaternionFromMatrix(const double *m, double *q) {
        double t = m[0] + m[4] + m[8];
        if (t > 0) {
            t = std::sqrt(t + 1.0);
            q[3] = 0.5 * t;
            t = 0.5 / t;
            q[0] = (m[7] - m[5]) * t;
            q[1] = (m[2] - m[6]) * t;
            q[2] = (m[3] - m[1]) * t;
        } else {
            int i = 0;
            if (m[4] > m[0]) i = 1;
            if (m[8] > m[4 * i]) i = 2;
            const int j = (i + 1) % 3, k = (j + 1) % 3;
            t = std::sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0);
            q[i] = 0.5 * t;
            t = 0.5 / t;
            q[3] = (m[3 * k + j] - m[3 * j + k]) * t;
            q[j] = (m[3 * j + i] + m[3 * i + j]) * t;
            q[k] = (m[3 * k + i] + m[3 * i + k]) * t;
        }
    }
    // keyframe pose as the reference stores it (EventCalibIni.cpp:270-278, identity body <- sensor extrinsics)
    static Frame makeFrame(double timeStamp, const EventCalibIni::FramePose &p, const std::vector<CirclesEventFrame::CalibCircle> &features,
                           const std::vector<int> &featureLandmark, size_t n_circles) {
        Frame f;
        f.timeStamp = timeStamp;
        for (int k = 0; k < 3; k++) f.twb[k] = p.twb[k];
        double Rwb[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) Rwb[3 * i + j] = p.Rsw[3 * j + i];
        quaternionFromMatrix(Rwb, f.Qwb);
        f.circles.assign(3 * n_circles, std::nan(""));
        for (size_t i = 0; i < features.size(); i++) {
            const size_t lm = featureLandmark.empty() ? i : (size_t) featureLandmark[i];
            f.circles[3 * lm] = features[i].location[0];
            f.circles[3 * lm + 1] = features[i].location[1];
            f.circles[3 * lm + 2] = features[i].radius;
        }
        return f;
    }

    EventCalibSpline(std::vector<Frame> frames, EventContainer::Ptr eventContainer, CirclePatternParameters::Ptr pattern, bool useSO3,
                     double motionTimeStep, const double K[4], const double distCoeffs[5], int maxIterations = 50,
                     bool fisheye = false)
        : frames_(std::move(frames)), eventContainer_(std::move(eventContainer)), pattern_(std::move(pattern)), useSO3_(useSO3),
          fisheye_(fisheye), motionTimeStep_(motionTimeStep), circleRadius_(pattern_->circleRadius), maxIterations_(maxIterations) {
        if (frames_.size() <= 10) throw std::logic_error("too few frames in the map.");  // :26-28
        std::sort(frames_.begin(), frames_.end(), [](const Frame &a, const Frame &b) { return a.timeStamp < b.timeStamp; });
        reduceMap();
        // spline per segment (:61-91)
        segCpOff_.assign(1, 0);
        for (const auto &seg : segments_) {
            std::vector<double> u, tw, qw;
            for (size_t id : seg) {
                u.push_back(frames_[id].timeStamp);
                tw.insert(tw.end(), frames_[id].twb, frames_[id].twb + 3);
                qw.insert(qw.end(), frames_[id].Qwb, frames_[id].Qwb + 4);
            }
            u.front() -= 3 * motionTimeStep_;  // spline can only evaluate inside the bound
            u.back() += 3 * motionTimeStep_;
            time2splineIdx_.emplace_back(u.front(), u.back());
            int cpNum = (int) std::floor((u.back() - u.front()) / (50 * motionTimeStep_));
            if (cpNum > (int) u.size()) cpNum = (int) u.size() - 1;
            if (cpNum < 4) cpNum = 4;  // becomes a Bezier curve
            std::vector<double> kn((size_t) cpNum + 4), ct((size_t) cpNum * 3), cq((size_t) cpNum * 4), kn2((size_t) cpNum + 4);
            int rc = ecal_spline_fit(u.data(), tw.data(), (uint32_t) u.size(), 3, (uint32_t) cpNum, kn.data(), ct.data());
            if (rc == ECAL_OK) rc = ecal_spline_fit(u.data(), qw.data(), (uint32_t) u.size(), 4, (uint32_t) cpNum, kn2.data(), cq.data());
            if (rc != ECAL_OK) throw std::logic_error(std::string("spline initialisation failed: ") + ecal_strerror(rc));
            if (useSO3_) {  // BsplineSO3::initialGuess stores unit quaternions (Sophus::SO3d::setQuaternion) ...
                for (int c = 0; c < cpNum; c++) {
                    double *q = &cq[4 * (size_t) c];
                    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
                    for (int k = 0; k < 4; k++) q[k] /= n;
                }
                // ... and the constructor ends with optimizeCP (BsplineSO3.cpp:55, :285-341): the fit on the group
                rc = ecal_spline_so3_refine(kn2.data(), (uint32_t) cpNum, cq.data(), qw.data(), u.data(), (uint32_t) u.size(), 0, nullptr,
                                            nullptr, nullptr);
                if (rc != ECAL_OK) throw std::logic_error(std::string("SO3 spline refinement failed: ") + ecal_strerror(rc));
            }
            knots_.insert(knots_.end(), kn.begin(), kn.end());
            cpQ_.insert(cpQ_.end(), cq.begin(), cq.end());
            cpT_.insert(cpT_.end(), ct.begin(), ct.end());
            segCpOff_.push_back(segCpOff_.back() + (uint32_t) cpNum);
        }
        // intrinsics (:93-105): only radial distortion (the reference throws for anything else, :97-99).  fisheye (new, BASELINE
        // configs[4]): distCoeffs = cv::fisheye's k1..k4, reverted by the same series into the solver's inverse polynomial
        const double radial[4] = {distCoeffs[0], distCoeffs[1], fisheye_ ? distCoeffs[2] : distCoeffs[4], fisheye_ ? distCoeffs[3] : 0.0};
        double inv[5];
        ecal_inverse_radial_distortion(radial, inv);
        for (int k = 0; k < 4; k++) intrinsics_[k] = K[k];
        for (int k = 0; k < 5; k++) intrinsics_[4 + k] = inv[k];
        optimize();
        updateMap();
    }

    const double *intrinsics() const { return intrinsics_; }  // fx fy cx cy k1..k5 (inverse radial polynomial)
    const std::vector<Frame> &frames() const { return frames_; }
    const Summary &summary() const { return summary_; }
    size_t splineNum() const { return segments_.size(); }

    int time2splineIdx(double t) const {
        for (size_t i = 0; i < time2splineIdx_.size(); i++)
            if (t >= time2splineIdx_[i].first && t <= time2splineIdx_[i].second) return (int) i;
        return -1;
    }

    // timestamp tx ty tz qx qy qz qw, one keyframe per line (identity body <- sensor extrinsics: Twc = Twb)
    void saveKeyFrameTrajectoryTUM(const std::string &filename) const {
        std::ofstream f(filename.c_str());
        f << std::fixed;
        for (const auto &bf : frames_)
            f << std::setprecision(10) << bf.timeStamp << " " << bf.twb[0] << " " << bf.twb[1] << " " << bf.twb[2] << " " << bf.Qwb[0]
              << " " << bf.Qwb[1] << " " << bf.Qwb[2] << " " << bf.Qwb[3] << std::endl;
    }

private:
    // :318-345 — a gap of more than 50 steps starts a new spline; segments with fewer than degree + 1 frames are dropped
    void reduceMap() {
        std::vector<std::vector<size_t>> sets(1);
        double last = frames_.front().timeStamp;
        for (size_t i = 0; i < frames_.size(); i++) {
            if (frames_[i].timeStamp - last > 50 * motionTimeStep_) sets.emplace_back();
            sets.back().push_back(i);
            last = frames_[i].timeStamp;
        }
        std::vector<Frame> kept;
        for (auto &s : sets) {
            if (s.size() < 4) continue;
            segments_.emplace_back();
            for (size_t id : s) {
                segments_.back().push_back(kept.size());
                kept.push_back(std::move(frames_[id]));
            }
        }
        frames_ = std::move(kept);
        if (segments_.empty()) throw std::logic_error("sampleSets not filtered");
    }

    bool optimize() {
        ecal_ctx *ctx = ecal_host::thread_ctx();
        const ecal_stream *es = eventContainer_->device();
        const uint64_t n_events = ecal_stream_size(es);
        const uint32_t n_circ = (uint32_t) (pattern_->rows * pattern_->cols), F = (uint32_t) frames_.size();
        std::vector<double> kf_time(F), kf_circ((size_t) F * n_circ * 3);
        for (uint32_t f = 0; f < F; f++) {
            kf_time[f] = frames_[f].timeStamp;
            std::copy(frames_[f].circles.begin(), frames_[f].circles.end(), kf_circ.begin() + (size_t) f * n_circ * 3);
        }
        std::vector<std::array<double, 3>> corners;
        std::vector<double> landmarks;
        for (int i = 0; i < pattern_->rows; i++)
            for (int j = 0; j < pattern_->cols; j++) {
                landmarks.push_back((pattern_->isAsymmetric ? (2 * j + i % 2) : j) * pattern_->squareSize);
                landmarks.push_back(i * pattern_->squareSize);
                landmarks.push_back(0.0);
            }
        ecal_spline_problem prob;
        prob.n_segments = (uint32_t) segments_.size();
        prob.seg_cp_off = segCpOff_.data();
        prob.knots = knots_.data();
        prob.n_res = 0;          // the residual arrays are made on the device (below)
        prob.obs = nullptr;
        prob.time = nullptr;
        prob.lm_id = nullptr;
        prob.seg_id = nullptr;
        prob.n_landmarks = n_circ;
        prob.landmarks = landmarks.data();
        prob.circle_radius = circleRadius_;
        prob.huber_a = 0.2 * circleRadius_;  // :205
        prob.use_so3 = useSO3_ ? 1 : 0;
        prob.camera_model = fisheye_ ? ECAL_CAMERA_FISHEYE : ECAL_CAMERA_RADIAL;
        // :158-192 — every event of every spline's range against its nearest keyframe's circles, and the Ceres problem's
        // residual blocks added where they are found (:181-235): one pass over the resident stream, the records stay in HBM
        std::vector<double> ranges;
        for (size_t i = 0; i < segments_.size(); i++) {
            ranges.push_back(time2splineIdx_[i].first);
            ranges.push_back(time2splineIdx_[i].second);
        }
        ecal_solver *solver = nullptr;
        int rc = ecal_solver_create_from_stream(ctx, es, kf_time.data(), kf_circ.data(), F, n_circ, ranges.data(), (uint32_t) segments_.size(),
                                                5 * motionTimeStep_, 5.0, &prob, &solver);
        if (rc != ECAL_OK) throw std::runtime_error(std::string("ecal_solver_create_from_stream: ") + ecal_last_error(ctx));
        (void) n_events;
        eventContainer_->release();  // eventContainer_->container.clear() (:194)
        summary_.residuals = ecal_solver_num_residuals(solver);
        if (summary_.residuals == 0) {
            ecal_solver_destroy(solver);
            return false;
        }
        std::vector<double> x(ecal_solver_param_size(solver));
        std::copy(intrinsics_, intrinsics_ + 9, x.begin());
        std::copy(cpQ_.begin(), cpQ_.end(), x.begin() + 9);
        std::copy(cpT_.begin(), cpT_.end(), x.begin() + 9 + cpQ_.size());
        ecal_lm_options opt;
        ecal_lm_default_options(&opt);  // the options of :238-243
        opt.max_num_iterations = maxIterations_;
        ecal_lm_summary sm;
        rc = ecal_solver_solve(solver, x.data(), &opt, &sm);
        ecal_solver_destroy(solver);
        if (rc != ECAL_OK) throw std::runtime_error(std::string("ecal_solver_solve: ") + ecal_last_error(ctx));
        std::copy(x.begin(), x.begin() + 9, intrinsics_);
        std::copy(x.begin() + 9, x.begin() + 9 + cpQ_.size(), cpQ_.begin());
        std::copy(x.begin() + 9 + cpQ_.size(), x.end(), cpT_.begin());
        summary_.iterations = sm.iterations;
        summary_.initial_cost = sm.initial_cost;
        summary_.final_cost = sm.final_cost;
        return true;
    }

    // :253-317 — keyframe poses re-read from the optimised splines (quaternion variant: blend + normalise; the SO3
    // variant's cumulative evaluation differs from the blend by O(curvature^2) between keyframes and is not replayed here)
    void updateMap() {
        for (auto &bf : frames_) {
            const int idx = time2splineIdx(bf.timeStamp);
            if (idx < 0) continue;
            const uint32_t n_cp = segCpOff_[idx + 1] - segCpOff_[idx];
            const double *kn = &knots_[segCpOff_[idx] + 4 * (size_t) idx];
            double t[3], q[4];
            if (ecal_spline_eval(kn, &cpT_[3 * (size_t) segCpOff_[idx]], n_cp, 3, &bf.timeStamp, 1, t) != ECAL_OK) continue;
            if (ecal_spline_eval(kn, &cpQ_[4 * (size_t) segCpOff_[idx]], n_cp, 4, &bf.timeStamp, 1, q) != ECAL_OK) continue;
            const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
            for (int k = 0; k < 3; k++) bf.twb[k] = t[k];
            for (int k = 0; k < 4; k++) bf.Qwb[k] = q[k] / n;
        }
    }

    std::vector<Frame> frames_;
    EventContainer::Ptr eventContainer_;
    CirclePatternParameters::Ptr pattern_;
    bool useSO3_, fisheye_;
    double motionTimeStep_, circleRadius_;
    int maxIterations_;
    std::vector<std::vector<size_t>> segments_;             // sampleIdSets_
    std::vector<std::pair<double, double>> time2splineIdx_;
    std::vector<uint32_t> segCpOff_;
    std::vector<double> knots_, cpQ_, cpT_;
    double intrinsics_[9];
    Summary summary_;
};

}  // namespace opengv2

#endif  // ECAL_HOST_EVENT_CALIB_SPLINE_HPP_
