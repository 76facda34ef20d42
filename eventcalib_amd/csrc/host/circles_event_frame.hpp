// Counterpart of the reference's CirclesEventFrame
// (event_camera_calib/include/opengv2/event_camera_calib/CirclesEventFrame.hpp,
//  event_camera_calib/src/CirclesEventFrame.cpp) on top of libecal.so.
//
// extractFeatures() runs the reference's extraction up to the candidate circles on the GPU.  The
// last step of the reference — cv::findCirclesGrid ordering the candidates into the pattern grid
// (:332-353) — is not part of this round: candidates() exposes the candidate list, and
// extractFeatures() returns whether the frame reached a full candidate set (>= rows*cols).
#ifndef ECAL_HOST_CIRCLES_EVENT_FRAME_HPP_
#define ECAL_HOST_CIRCLES_EVENT_FRAME_HPP_

#include "event.hpp"

namespace opengv2 {

struct CirclePatternParameters {
    typedef std::shared_ptr<CirclePatternParameters> Ptr;
    bool isAsymmetric = true;
    int rows = 9, cols = 4;
    double squareSize = 5.5, circleRadius = 1.75;
};

class CirclesEventFrame : public EventFrame {
public:
    struct Params {
        Params() : dbscan_eps(4), dbscan_startMinSample(2), clusterMinSample(5), knn_num(3), fitCircle(false) {}
        double dbscan_eps;          // pixel unit
        int dbscan_startMinSample;
        int clusterMinSample;
        int knn_num;
        bool fitCircle;             // false: midpoint circles (:283-311); true: algebraic fit + knn (:180-281)
    };

    CirclesEventFrame(EventContainer::Ptr container, const std::pair<double, double> &duration,
                      CirclePatternParameters::Ptr pattern, Params params = Params())
        : EventFrame(std::move(container), duration), pattern_(std::move(pattern)), params_(params) {
        circleRadiusThreshold_ = ecal_circle_radius_threshold(container_->cameraSize[0], container_->cameraSize[1],
                                                              pattern_->rows, pattern_->cols, pattern_->isAsymmetric,
                                                              pattern_->squareSize, pattern_->circleRadius);
    }

    // false when a polarity is empty, when fewer than rows*cols clusters survive the size filter
    // in either polarity (CirclesEventFrame.cpp:62-64,127-129), or when fewer than rows*cols
    // candidate circles are found (the reference's findCirclesGrid cannot succeed with fewer).
    bool extractFeatures() {
        ensure();
        if (det_.status != 0) return false;
        return det_.candidates.size() >= (size_t) (pattern_->rows * pattern_->cols);
    }

    const FrameDetection &detection() {
        ensure();
        return det_;
    }
    double circleRadiusThreshold() const { return circleRadiusThreshold_; }

protected:
    ecal_detect_params params() const override {
        ecal_detect_params p;
        p.dbscan_eps = params_.dbscan_eps;
        p.dbscan_min_samples = (uint32_t) params_.dbscan_startMinSample;
        p.cluster_min_sample = (uint32_t) params_.clusterMinSample;
        p.need_clusters = (uint32_t) (pattern_->rows * pattern_->cols);
        p.circle_radius_threshold = circleRadiusThreshold_;
        p.fit_circle = params_.fitCircle ? 1 : 0;
        p.knn_num = (uint32_t) params_.knn_num;
        return p;
    }
    CirclePatternParameters::Ptr pattern_;
    Params params_;
    double circleRadiusThreshold_;
};

}  // namespace opengv2

#endif  // ECAL_HOST_CIRCLES_EVENT_FRAME_HPP_
