// Counterpart of the reference's CirclesEventFrame
// (event_camera_calib/include/opengv2/event_camera_calib/CirclesEventFrame.hpp,
//  event_camera_calib/src/CirclesEventFrame.cpp) on top of libecal.so.
//
// extractFeatures() runs the whole reference function on the GPU: DBSCAN x2, cluster filter, medians,
// pairing / circle fit, and the ordering of the candidates into the pattern grid (a deterministic lattice
// walk in place of the vendored cv::findCirclesGrid, ecal_grid_order_dev); features() are the circles in
// grid order, index i*cols + j <-> landmark ((2j + i%2) s, i s, 0).
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

    struct CalibCircle {  // event_camera_calib/include/opengv2/event_camera_calib/CalibCircle.hpp
        Vector2d location;
        double radius;
    };

    // false when a polarity is empty, when fewer than rows*cols clusters survive the size filter in either
    // polarity (CirclesEventFrame.cpp:62-64,127-129), or when no complete grid is found among the candidates
    // (findCirclesGrid's isFound, :332-336,358).
    bool extractFeatures() {
        ensure();
        features_.clear();
        if (det_.status != 0 || !det_.gridFound) return false;
        for (size_t idx : det_.orderIdxs)  // :350-353
            features_.push_back(CalibCircle{det_.candidateCenters[idx], det_.candidatesRadius[idx]});
        return true;
    }
    const std::vector<CalibCircle> &features() const { return features_; }

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
        p.rows = (uint32_t) pattern_->rows;
        p.cols = (uint32_t) pattern_->cols;
        return p;
    }
    CirclePatternParameters::Ptr pattern_;
    Params params_;
    double circleRadiusThreshold_;
    std::vector<CalibCircle> features_;
};

}  // namespace opengv2

#endif  // ECAL_HOST_CIRCLES_EVENT_FRAME_HPP_
