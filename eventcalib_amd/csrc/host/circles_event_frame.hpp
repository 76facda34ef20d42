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

#include <unordered_set>

#include "event.hpp"
#include "file_settings.hpp"

namespace opengv2 {

struct CirclePatternParameters {
    typedef std::shared_ptr<CirclePatternParameters> Ptr;
    CirclePatternParameters() = default;
    explicit CirclePatternParameters(const FileSettings &node) {   // parameters.hpp:15-21
        node["BoardSize_Cols"] >> cols;
        node["BoardSize_Rows"] >> rows;
        node["Square_Size"] >> squareSize;
        node["Is_Pattern_Asymmetric"] >> isAsymmetric;
        node["Circles_Radius"] >> circleRadius;
    }
    bool isAsymmetric = true;
    int rows = 9, cols = 4;
    double squareSize = 5.5, circleRadius = 1.75;
};

class CirclesEventFrame : public EventFrame {
public:
    struct Params {
        Params() : dbscan_eps(4), dbscan_startMinSample(2), clusterMinSample(5), knn_num(3), fitCircle(false) {}
        explicit Params(const FileSettings &node) : Params() {   // CirclesEventFrame.cpp:42-48
            node["dbscan_eps"] >> dbscan_eps;
            node["dbscan_startMinSample"] >> dbscan_startMinSample;
            node["clusterMinSample"] >> clusterMinSample;
            node["knn_num"] >> knn_num;
            node["fitCircle"] >> fitCircle;
        }
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

    // CirclesEventFrame::image() as the reference's driver saves it (eventCameraCalib.cpp:214-227): black sensor-sized RGB
    // raster; the kept clusters' pixels in the reference's colours (BGR (c / 256, c % 256, 200 | 100), c = 20 x cluster index,
    // CirclesEventFrame.cpp:89-117); the candidate circles in green (:315-318) and the features of the grid in white (:626).
    // cv::circle's rasterisation and drawChessboardCorners are OpenCV's: the outlines here are midpoint circles, the grid's
    // row connections are left out — a diagnostic picture, not a parity item.
    std::vector<uint8_t> image() {
        ensure();
        const int w = (int) container_->cameraSize[0], h = (int) container_->cameraSize[1];
        std::vector<uint8_t> rgb((size_t) w * h * 3, 0);
        auto put = [&](int x, int y, uint8_t r, uint8_t g, uint8_t b) {
            if (x < 0 || y < 0 || x >= w || y >= h) return;
            uint8_t *p = &rgb[3 * ((size_t) y * w + x)];
            p[0] = r, p[1] = g, p[2] = b;
        };
        auto ring = [&](double cx, double cy, double rad, uint8_t r, uint8_t g, uint8_t b) {
            const int x0 = (int) cx, y0 = (int) cy, R = (int) rad;   // cv::Point(double, double) / int radius truncate
            int x = R, y = 0, err = 1 - R;
            while (x >= y) {
                const int px[8] = {x, y, -y, -x, -x, -y, y, x}, py[8] = {y, x, x, y, -y, -x, -x, -y};
                for (int k = 0; k < 8; k++) put(x0 + px[k], y0 + py[k], r, g, b);
                y++;
                if (err < 0) err += 2 * y + 1;
                else {
                    x--;
                    err += 2 * (y - x) + 1;
                }
            }
        };
        for (size_t i = 0; i < det_.positive.size(); i++)
            if (i < det_.keptPos.size() && det_.keptPos[i] >= 0) {
                const unsigned c = 20u * (unsigned) det_.keptPos[i];
                put((int) det_.positive[i][0], (int) det_.positive[i][1], 200, (uint8_t) (c % 256), (uint8_t) (c / 256));   // BGR (c/256, c%256, 200)
            }
        for (size_t i = 0; i < det_.negative.size(); i++)
            if (i < det_.keptNeg.size() && det_.keptNeg[i] >= 0) {
                const unsigned c = 20u * (unsigned) det_.keptNeg[i];
                put((int) det_.negative[i][0], (int) det_.negative[i][1], 100, (uint8_t) (c % 256), (uint8_t) (c / 256));
            }
        for (size_t i = 0; i < det_.candidateCenters.size(); i++)
            ring(det_.candidateCenters[i][0], det_.candidateCenters[i][1], det_.candidatesRadius[i], 0, 255, 0);
        for (const CalibCircle &f : features_) ring(f.location[0], f.location[1], f.radius, 255, 255, 255);
        return rgb;
    }

    // camera model rectifyFeatures projects with (the reference reads it from sensor_: PinholeCamera K, distCoeffs
    // k1 k2 p1 p2 k3 and size(), CirclesEventFrame.cpp:426-433)
    struct Camera {
        double fx, fy, cx, cy;
        double distCoeffs[5];
        bool fisheye = false;   // distCoeffs[0..3] = cv::fisheye's k1..k4 (new: BASELINE configs[4])
    };

    // CirclesEventFrame::rectifyFeatures (:417-638) after extractFeatures() succeeded: every feature is re-found
    // around the projection of its landmark under the pose (Rcw row-major, tcw); erased features leave
    // features_ (their grid index is kept in featureLandmark()), false = discard the frame.  `outlierIdxs` is
    // unused, as in the reference.
    bool rectifyFeatures(const std::unordered_set<int> & /*outlierIdxs*/, const double (&Rcw)[9], const double (&tcw)[3],
                         const Camera &camera) {
        ensure();
        const uint32_t n = (uint32_t) (pattern_->rows * pattern_->cols);
        if (features_.size() != n) return false;
        std::vector<double> xy, lm(3 * (size_t) n), feat(3 * (size_t) n);
        std::vector<int32_t> kept;
        for (const auto &p : det_.positive) xy.push_back(p[0]), xy.push_back(p[1]);
        for (const auto &p : det_.negative) xy.push_back(p[0]), xy.push_back(p[1]);
        kept.insert(kept.end(), det_.keptPos.begin(), det_.keptPos.end());
        kept.insert(kept.end(), det_.keptNeg.begin(), det_.keptNeg.end());
        const uint32_t off[2] = {0, (uint32_t) det_.positive.size()};
        const uint32_t cnt[2] = {(uint32_t) det_.positive.size(), (uint32_t) det_.negative.size()};
        for (int i = 0; i < pattern_->rows; i++)      // EventCalibIni::calcBoardCornerPositions (:98-112), cv::Point3f
            for (int j = 0; j < pattern_->cols; j++) {
                double *o = &lm[3 * (size_t) (i * pattern_->cols + j)];
                o[0] = (float) ((pattern_->isAsymmetric ? (2 * j + i % 2) : j) * pattern_->squareSize);
                o[1] = (float) (i * pattern_->squareSize);
                o[2] = 0;
            }
        double pose[12];
        for (int i = 0; i < 9; i++) pose[i] = Rcw[i];
        for (int i = 0; i < 3; i++) pose[9 + i] = tcw[i];
        ecal_rectify_params prm;
        prm.fx = camera.fx, prm.fy = camera.fy, prm.cx = camera.cx, prm.cy = camera.cy;
        for (int i = 0; i < 5; i++) prm.dist[i] = camera.distCoeffs[i];
        prm.width = container_->cameraSize[0], prm.height = container_->cameraSize[1];
        prm.rows = (uint32_t) pattern_->rows, prm.cols = (uint32_t) pattern_->cols;
        prm.asymmetric = pattern_->isAsymmetric ? 1 : 0;
        prm.circle_radius = pattern_->circleRadius;
        prm.fit_circle = params_.fitCircle ? 1 : 0;
        prm.model = camera.fisheye ? 1 : 0;
        std::vector<uint32_t> valid(n);
        uint32_t info[2] = {0, 0};
        const int rc = ecal_rectify_batch(ecal_host::thread_ctx(), xy.data(), off, cnt, kept.data(),
                                          (uint32_t) kept.size(), pose, 1, lm.data(), &prm, feat.data(), valid.data(), info);
        if (rc != ECAL_OK)
            throw std::runtime_error(std::string("ecal_rectify_batch: ") + ecal_last_error(ecal_host::thread_ctx()));
        features_.clear();
        featureLandmark_.clear();
        for (uint32_t k = 0; k < n; k++)
            if (valid[k]) {
                features_.push_back(CalibCircle{Vector2d{{feat[3 * k], feat[3 * k + 1]}}, feat[3 * k + 2]});
                featureLandmark_.push_back((int) k);
            }
        return info[0] != 0;
    }
    // landmark (grid) index of features()[i] after rectifyFeatures (the reference keeps f->landmark())
    const std::vector<int> &featureLandmark() const { return featureLandmark_; }

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
    std::vector<int> featureLandmark_;
};

}  // namespace opengv2

#endif  // ECAL_HOST_CIRCLES_EVENT_FRAME_HPP_
