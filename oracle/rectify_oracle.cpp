// TEST INFRASTRUCTURE — CPU restatement of CirclesEventFrame::rectifyFeatures
// (event_camera_calib/src/CirclesEventFrame.cpp:417-638).  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may use anything under oracle/.
//
// Third-party arithmetic on this path, absent from /root/reference and restated from the published algorithms:
//  * cv::projectPoints (OpenCV >= 4.0 calib3d, CMakeLists.txt:43; call site :449): Point3f object points widened
//    to double, X = R P + t, pinhole division, radial (k1,k2,k3) + tangential (p1,p2) distortion, K, result
//    narrowed to Point2f.  The reference passes rvec = Rodrigues(Rcw) and OpenCV converts it back; that round
//    trip (identity to ~1e-16) is not replayed: R is used as given.
//  * nanoflann radiusSearch (1.3.x RadiusResultSet::addPoint: `dist < radius`, L2_Simple = plain sum of squared
//    differences): the set of points strictly inside the squared radius; the result order does not matter here.
// No reference test pins this function => parity unpinned; the GPU kernel is compared with this restatement.
//
// Deliberate, documented difference: fitCircle's nine sums run over the + members then the - members in
// ascending pid (reference: ascending cluster id, BFS member order inside a cluster).  Event pixels are integers
// below 2^14, all nine sums are exact, so the order cannot change the result on event data.
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <vector>

extern "C" int oracle_fit_circle(const double *a_xy, uint32_t na, const double *b_xy, uint32_t nb, double *centre_xy,
                                 double *radius);

namespace {

// cv::projectPoints, one point (no rational / thin-prism / tilt terms: the reference has 5 coefficients)
int g_fisheye = 0;   // set by oracle_rectify_cam around its call (the oracle is single-threaded test infrastructure)
void project_point(const double *R, const double *t, const double *K4, const double *k, float X, float Y, float Z,
                   float *u, float *v) {
    const double Xd = X, Yd = Y, Zd = Z;
    double x = R[0] * Xd + R[1] * Yd + R[2] * Zd + t[0];
    double y = R[3] * Xd + R[4] * Yd + R[5] * Zd + t[1];
    double z = R[6] * Xd + R[7] * Yd + R[8] * Zd + t[2];
    z = z ? 1. / z : 1;
    x *= z;
    y *= z;
    if (g_fisheye) {   // cv::fisheye::projectPoints (k = k1..k4, alpha = 0): the build's fisheye chain (BASELINE configs[4], new)
        const double r = std::sqrt(x * x + y * y), th = std::atan(r), th2 = th * th;
        const double thd = th * (1 + th2 * (k[0] + th2 * (k[1] + th2 * (k[2] + th2 * k[3]))));
        const double sc = r > 1e-8 ? thd / r : 1.0;
        *u = (float) (x * sc * K4[0] + K4[2]);
        *v = (float) (y * sc * K4[1] + K4[3]);
        return;
    }
    const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
    const double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
    const double cdist = 1 + k[0] * r2 + k[1] * r4 + k[4] * r6;
    const double xd = x * cdist + k[2] * a1 + k[3] * a2;
    const double yd = y * cdist + k[2] * a3 + k[3] * a1;
    *u = (float) (xd * K4[0] + K4[2]);
    *v = (float) (yd * K4[1] + K4[3]);
}

}  // namespace

extern "C" {

// One keyframe.  pos_xy/neg_xy: positiveEvents_/negativeEvents_; kept_pos/kept_neg: index of the point's cluster in
// pClusters_/nClusters_ (:120-121) or -1.  pose = Rcw row-major (9) + tcw (3); camera = fx fy cx cy; dist = k1 k2
// p1 p2 k3; landmarks [n][3] in grid order (n = rows*cols).  Outputs: feat_xyr [n][3], feat_valid [n] (0 = erased),
// info[2] = {return value, erased count}.
int oracle_rectify(const double *pos_xy, uint32_t n_pos, const double *neg_xy, uint32_t n_neg, const int32_t *kept_pos,
                   const int32_t *kept_neg, const double *pose, const double *camera, const double *dist, double width,
                   double height, const double *landmarks, uint32_t rows, uint32_t cols, int asymmetric,
                   double circle_radius, int fit_circle, double *feat_xyr, uint32_t *feat_valid, uint32_t *info);
// model 1: the projections of rectifyFeatures go through cv::fisheye::projectPoints (dist[0..3] = k1..k4)
int oracle_rectify_cam(const double *pos_xy, uint32_t n_pos, const double *neg_xy, uint32_t n_neg, const int32_t *kept_pos,
                       const int32_t *kept_neg, const double *pose, const double *camera, const double *dist, double width,
                       double height, const double *landmarks, uint32_t rows, uint32_t cols, int asymmetric,
                       double circle_radius, int fit_circle, double *feat_xyr, uint32_t *feat_valid, uint32_t *info, int model) {
    g_fisheye = model == 1;
    const int rc = oracle_rectify(pos_xy, n_pos, neg_xy, n_neg, kept_pos, kept_neg, pose, camera, dist, width, height, landmarks, rows, cols,
                                  asymmetric, circle_radius, fit_circle, feat_xyr, feat_valid, info);
    g_fisheye = 0;
    return rc;
}
int oracle_rectify(const double *pos_xy, uint32_t n_pos, const double *neg_xy, uint32_t n_neg, const int32_t *kept_pos,
                   const int32_t *kept_neg, const double *pose, const double *camera, const double *dist, double width,
                   double height, const double *landmarks, uint32_t rows, uint32_t cols, int asymmetric,
                   double circle_radius, int fit_circle, double *feat_xyr, uint32_t *feat_valid, uint32_t *info) {
    const uint32_t n = rows * cols;
    const double *R = pose, *t = pose + 9;
    const double *xy[2] = {pos_xy, neg_xy};
    const uint32_t cnt[2] = {n_pos, n_neg};
    const int32_t *kept[2] = {kept_pos, kept_neg};
    int32_t n_clusters[2] = {0, 0};
    for (int s = 0; s < 2; s++)
        for (uint32_t i = 0; i < cnt[s]; i++) n_clusters[s] = std::max(n_clusters[s], kept[s][i] + 1);

    for (uint32_t k = 0; k < n; k++) {
        feat_valid[k] = 0;
        feat_xyr[3 * k] = feat_xyr[3 * k + 1] = feat_xyr[3 * k + 2] = std::nan("");
        const double *c = landmarks + 3 * k;
        const double skew = circle_radius / std::sqrt(2);  // :438
        const float obj[5][3] = {{(float) c[0], (float) c[1], (float) c[2]},
                                 {(float) (c[0] + skew), (float) (c[1] + skew), (float) c[2]},
                                 {(float) (c[0] + skew), (float) (c[1] - skew), (float) c[2]},
                                 {(float) (c[0] - skew), (float) (c[1] - skew), (float) c[2]},
                                 {(float) (c[0] - skew), (float) (c[1] + skew), (float) c[2]}};
        double img[5][2];
        float u0 = 0, v0 = 0;
        for (int i = 0; i < 5; i++) {
            float u, v;
            project_point(R, t, camera, dist, obj[i][0], obj[i][1], obj[i][2], &u, &v);
            img[i][0] = u;
            img[i][1] = v;
            if (i == 0) u0 = u, v0 = v;
        }
        if (u0 >= width || v0 >= height || u0 < 0 || v0 < 0) continue;  // :457-461

        double radius[4], max_radius = 0;
        for (int i = 1; i < 5; i++) {
            const double dx = img[i][0] - img[0][0], dy = img[i][1] - img[0][1];
            radius[i - 1] = std::sqrt(dx * dx + dy * dy);
            if (radius[i - 1] > max_radius) max_radius = radius[i - 1];
        }
        const double inlier = 3;
        const double search = (max_radius + inlier) * (max_radius + inlier);  // std::pow(.,2), :474

        // inliers (:483-520) -> the clusters they belong to (:523-545)
        std::vector<uint8_t> sel[2];
        for (int s = 0; s < 2; s++) {
            sel[s].assign(n_clusters[s], 0);
            for (uint32_t i = 0; i < cnt[s]; i++) {
                const double dx = xy[s][2 * i] - img[0][0], dy = xy[s][2 * i + 1] - img[0][1];
                const double d2 = dx * dx + dy * dy;
                if (!(d2 < search)) continue;
                const double distance = std::sqrt(d2);
                int idx = 0;
                if (dx >= 0 && dy >= 0) idx = 0;
                else if (dx >= 0 && dy <= 0) idx = 1;
                else if (dx <= 0 && dy <= 0) idx = 2;
                else if (dx <= 0 && dy >= 0) idx = 3;
                if (std::abs(distance - radius[idx]) <= inlier && kept[s][i] >= 0) sel[s][kept[s][i]] = 1;
            }
        }
        // whole clusters (:546-557)
        std::vector<double> members[2];
        for (int s = 0; s < 2; s++)
            for (uint32_t i = 0; i < cnt[s]; i++)
                if (kept[s][i] >= 0 && sel[s][kept[s][i]]) {
                    members[s].push_back(xy[s][2 * i]);
                    members[s].push_back(xy[s][2 * i + 1]);
                }
        if (members[0].size() / 2 < 5 || members[1].size() / 2 < 5) continue;  // :560-563

        double centre[2], r;
        oracle_fit_circle(members[0].data(), (uint32_t) (members[0].size() / 2), members[1].data(),
                          (uint32_t) (members[1].size() / 2), centre, &r);
        std::nth_element(radius, radius + 2, radius + 4);  // :570
        const double ex = centre[0] - img[0][0], ey = centre[1] - img[0][1];
        if (std::sqrt(ex * ex + ey * ey) > 2 * inlier || std::abs(r - radius[2]) > 1.5 * inlier) continue;  // :572-576

        feat_valid[k] = 1;
        feat_xyr[3 * k] = centre[0];
        feat_xyr[3 * k + 1] = centre[1];
        feat_xyr[3 * k + 2] = r;
    }

    // erased features on the pattern's border count more (:587-622)
    const int step = (asymmetric ? 2 : 1) * (int) cols, total = (int) n;
    std::vector<std::vector<int>> edge(4);
    for (int i = 0; i < (int) cols; i++) edge[0].push_back(i);
    for (int i = ((int) rows - 1) * (int) cols; i < total; i++) edge[1].push_back(i);
    for (int i = 0; i < total; i += step) edge[2].push_back(i);
    for (int i = asymmetric ? 2 * (int) cols - 1 : (int) cols - 1; i < total; i += step) edge[3].push_back(i);
    int score[4] = {0, 0, 0, 0}, erased = 0;
    for (int k = 0; k < total; k++)
        if (!feat_valid[k]) {
            erased++;
            for (int e = 0; e < 4; e++)
                if (std::find(edge[e].begin(), edge[e].end(), k) != edge[e].end()) score[e]++;
        }
    info[1] = (uint32_t) erased;
    info[0] = 1;
    if (!fit_circle)
        for (int e = 0; e < 4; e++)
            if (score[e] >= (int) (edge[e].size() - 1)) info[0] = 0;
    if (erased >= 0.2 * (cols * rows)) info[0] = 0;  // :625-627
    return 0;
}

}  // extern "C"
