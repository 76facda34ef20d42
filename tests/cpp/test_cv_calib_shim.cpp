// cv::findCirclesGrid as the reference calls it (CirclesEventFrame.cpp:321-353), on the libecal shim host/cv_calib.hpp:
// reads candidate lists (n, then n x y pairs) from stdin until EOF; prints "found" + the 36 candidate indices the reference's
// nearest-candidate lookup (:340-353) gives for the returned centres, or "none".
#include <cstdio>
#include <limits>
#include "../../eventcalib_amd/csrc/host/cv_calib.hpp"

int main() {
    int n;
    while (std::scanf("%d", &n) == 1) {
        std::vector<cv::Point2f> points, outCenters;
        for (int i = 0; i < n; i++) {
            double x, y;
            if (std::scanf("%lf %lf", &x, &y) != 2) return 2;
            points.emplace_back((float) x, (float) y);
        }
        bool isFound = cv::findCirclesGrid(points, cv::Size(4, 9), outCenters, cv::CALIB_CB_ASYMMETRIC_GRID);
        if (!isFound) isFound = cv::findCirclesGrid(points, cv::Size(4, 9), outCenters, cv::CALIB_CB_ASYMMETRIC_GRID | cv::CALIB_CB_CLUSTERING);
        if (!isFound) {
            std::printf("none\n");
            continue;
        }
        std::printf("found");
        for (const auto &c : outCenters) {   // the caller's lookup: nearest candidate of every returned centre
            int best = -1;
            double bd = std::numeric_limits<double>::infinity();
            for (int i = 0; i < n; i++) {
                const double dx = points[i].x - c.x, dy = points[i].y - c.y;
                if (dx * dx + dy * dy < bd) {
                    bd = dx * dx + dy * dy;
                    best = i;
                }
            }
            std::printf(" %d", best);
        }
        std::printf("\n");
    }
    return 0;
}
