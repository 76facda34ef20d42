// Counterpart of the reference's vendored, patched circle-grid finder as its caller sees it:
//   bool cv::findCirclesGrid(const std::vector<Point2f> &points_, Size patternSize, OutputArray _centers, int flags,
//                            const CirclesGridFinderParameters &parameters_)
// (modules/camera_calibration/cv_calib/include/cv_calib.hpp:19-21, src/cv_calib.cpp:7-87, src/circlesgrid.cpp), called at
// event_camera_calib/src/CirclesEventFrame.cpp:332-336 with CALIB_CB_ASYMMETRIC_GRID and, after a failure, with
// CALIB_CB_ASYMMETRIC_GRID | CALIB_CB_CLUSTERING.  An unchanged CirclesEventFrame.cpp that includes THIS header in place of the
// vendored one links against libecal.so's ecal_grid_order (the deterministic lattice walk of ecal_grid.hip) and gets the centres
// in the same convention: row by row, index i * cols + j <-> model point ((2 j + i % 2) s, i s) (circlesgrid.cpp:1258-1291).
// With OpenCV present, define ECAL_CV_CALIB_USE_OPENCV before including: cv::Point2f / cv::Size / OutputArray are OpenCV's; without
// it (this image) the few types the signature needs are declared here.
#ifndef ECAL_HOST_CV_CALIB_HPP_
#define ECAL_HOST_CV_CALIB_HPP_

#include <vector>

#include "event.hpp"

#ifdef ECAL_CV_CALIB_USE_OPENCV
#include <opencv2/core.hpp>
#else
namespace cv {
struct Point2f {
    Point2f() : x(0), y(0) {}
    Point2f(float x_, float y_) : x(x_), y(y_) {}
    float x, y;
};
struct Size {
    Size() : width(0), height(0) {}
    Size(int w, int h) : width(w), height(h) {}
    int width, height;   // findCirclesGrid's patternSize = Size(points_per_row = cols, points_per_colum = rows)
};
enum { CALIB_CB_SYMMETRIC_GRID = 1, CALIB_CB_ASYMMETRIC_GRID = 2, CALIB_CB_CLUSTERING = 4 };
typedef std::vector<Point2f> &OutputArray;
struct CirclesGridFinderParameters {};   // (densityNeighborhoodSize, minDensity, kmeansAttempts, … of the vendored finder: not used)
}  // namespace cv
#endif

namespace cv {

// true: every centre found and ordered.  Only the asymmetric pattern is implemented (what the reference calls it with,
// parameters.hpp:15-21 Is_Pattern_Asymmetric: 1); CALIB_CB_CLUSTERING selects nothing here — the walk's robust starts are
// what the reference's clustering variant is for — so the caller's retry returns the first call's verdict.
inline bool findCirclesGrid(const std::vector<Point2f> &points_, Size patternSize, OutputArray _centers, int flags,
                            const CirclesGridFinderParameters & = CirclesGridFinderParameters()) {
    if (!(flags & CALIB_CB_ASYMMETRIC_GRID)) throw std::invalid_argument("cv::findCirclesGrid (libecal shim): asymmetric grids only");
    const uint32_t cols = (uint32_t) patternSize.width, rows = (uint32_t) patternSize.height, M = rows * cols;
    std::vector<double> xyr(3 * points_.size());
    for (size_t i = 0; i < points_.size(); i++) {
        xyr[3 * i] = points_[i].x;
        xyr[3 * i + 1] = points_[i].y;
        xyr[3 * i + 2] = 0.0;
    }
    std::vector<int32_t> order(M ? M : 1);
    uint32_t found = 0;
    ecal_ctx *ctx = ecal_host::thread_ctx();
    const int rc = ecal_grid_order(ctx, xyr.data(), (uint32_t) points_.size(), rows, cols, order.data(), &found);
    if (rc != ECAL_OK) throw std::runtime_error(std::string("ecal_grid_order: ") + ecal_strerror(rc) + " — " + ecal_last_error(ctx));
    std::vector<Point2f> centers;
    if (found)
        for (uint32_t m = 0; m < M; m++) centers.push_back(points_[(size_t) order[m]]);
#ifdef ECAL_CV_CALIB_USE_OPENCV
    Mat(centers).copyTo(_centers);
#else
    _centers = centers;
#endif
    return found != 0;
}

}  // namespace cv

#endif
