// Re-detection of the pattern circles around their predicted projections.
//
// Replaces CirclesEventFrame::rectifyFeatures (event_camera_calib/src/CirclesEventFrame.cpp:417-638) for a batch
// of keyframes: per circle, cv::projectPoints of the landmark and of four points on its rim (:428-455), the
// nanoflann radius searches over both polarities (:471-480), the quadrant-wise inlier test (:483-520), expansion
// of the inliers to the whole DBSCAN clusters they belong to (:523-557), CirclesEventFrame::fitCircle (:361-415)
// and the acceptance gates (:560-576); per keyframe the border score and the 20 % rule (:587-627).
//
// One wave per keyframe, lane k owns circle k (k + 64, ...): the per-circle work is a short sequential scan of
// the window's ~1.3 k points, all lanes read the same point (one broadcast load), and the per-lane set of
// selected clusters is a bit vector in LDS laid out word-major so the lanes never share a bank.  Sums are
// accumulated per lane in point order, which makes the result reproducible bit for bit (oracle/rectify_oracle.cpp
// documents the order and why it equals the reference's on event pixels).
//
// cv::projectPoints is third-party (OpenCV >= 4.0): restated from its published algorithm, float in / float out;
// the reference's Rodrigues round trip of Rcw is not replayed.
#include "ecal_ctx.hpp"

#pragma clang fp contract(off)

namespace ecal {

constexpr int RC_T = 64;
constexpr uint32_t RC_MAXK = 2048;            // kept clusters per polarity (ecal_extract_batch_dev's limit)
constexpr uint32_t RC_WORDS = RC_MAXK / 32;   // flag words per polarity and lane
constexpr uint32_t RC_MAXN = 128;             // circles per pattern

struct RectifyConst {
    double fx, fy, cx, cy;
    double k[5];
    double width, height;
    double circle_radius;
    uint32_t rows, cols;
    int asymmetric, fit_circle, model;
};

__device__ __forceinline__ void project_point(const double *R, const double *t, const RectifyConst &p, float X, float Y,
                                              float Z, double *u, double *v) {
    const double Xd = X, Yd = Y, Zd = Z;
    double x = R[0] * Xd + R[1] * Yd + R[2] * Zd + t[0];
    double y = R[3] * Xd + R[4] * Yd + R[5] * Zd + t[1];
    double z = R[6] * Xd + R[7] * Yd + R[8] * Zd + t[2];
    z = z != 0.0 ? 1. / z : 1;
    x *= z;
    y *= z;
    if (p.model == 1) {   // cv::fisheye::projectPoints (Kannala-Brandt k1..k4 = k[0..3], no skew): BASELINE configs[4]
        const double r = sqrt(x * x + y * y), th = atan(r), th2 = th * th;
        const double thd = th * (1 + th2 * (p.k[0] + th2 * (p.k[1] + th2 * (p.k[2] + th2 * p.k[3]))));
        const double sc = r > 1e-8 ? thd / r : 1.0;
        *u = (double) (float) (x * sc * p.fx + p.cx);
        *v = (double) (float) (y * sc * p.fy + p.cy);
        return;
    }
    const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
    const double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
    const double cdist = 1 + p.k[0] * r2 + p.k[1] * r4 + p.k[4] * r6;
    const double xd = x * cdist + p.k[2] * a1 + p.k[3] * a2;
    const double yd = y * cdist + p.k[2] * a3 + p.k[3] * a1;
    *u = (double) (float) (xd * p.fx + p.cx);
    *v = (double) (float) (yd * p.fy + p.cy);
}

// Eigen's A.lu().solve(b) for the 3x3 system of fitCircle: partial pivoting, same elimination as ecal_detect.hip
__device__ __forceinline__ void solve3(double (&A)[3][4], double (&x)[3]) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
        int piv = c;
#pragma unroll
        for (int r = c + 1; r < 3; r++)
            if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const double a = A[c][k], b = A[piv][k];
            A[c][k] = b;
            A[piv][k] = (piv == c) ? b : a;
        }
#pragma unroll
        for (int r = c + 1; r < 3; r++) {
            const double f = A[r][c] / A[c][c];
#pragma unroll
            for (int k = c; k < 4; k++) A[r][k] -= f * A[c][k];
        }
    }
    x[2] = A[2][3] / A[2][2];
    x[1] = (A[1][3] - A[1][2] * x[2]) / A[1][1];
    x[0] = (A[0][3] - A[0][1] * x[1] - A[0][2] * x[2]) / A[0][0];
}

// a point of the window as every lane needs it: lane j of the wave has loaded point i0 + j (one coalesced load for 64 points
// instead of a broadcast load per point), the walk reads them lane by lane (v_readlane: wave-uniform values)
__device__ __forceinline__ double rc_lane_f64(double v, uint32_t j) {
    const unsigned long long b = (unsigned long long) __double_as_longlong(v);
    const uint32_t lo = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) b, (int) j);
    const uint32_t hi = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (b >> 32), (int) j);
    return __longlong_as_double((long long) (((unsigned long long) hi << 32) | lo));
}

// WORDS = flag words per polarity and lane: the launch with RC_WORDS_SMALL (256 kept clusters per polarity: 4 KB of LDS, so that
// the compute unit holds waves enough to hide the walk's latencies) takes the frames that fit, the one with RC_WORDS (32 KB:
// five waves per compute unit) the others; a frame is worked on by exactly one of the two.
constexpr uint32_t RC_WORDS_SMALL = 8;
template <uint32_t WORDS>
__global__ __launch_bounds__(RC_T) void rectify_kernel(const double2 *__restrict__ xy,
                                                       const uint32_t *__restrict__ seg_off,
                                                       const uint32_t *__restrict__ seg_cnt,
                                                       const int32_t *__restrict__ kept_labels,
                                                       const uint32_t *__restrict__ win_info,
                                                       const uint32_t *__restrict__ frame_window,
                                                       const double *__restrict__ pose,
                                                       const double *__restrict__ landmarks, RectifyConst prm,
                                                       double *__restrict__ feat_xyr, uint32_t *__restrict__ feat_valid,
                                                       uint32_t *__restrict__ frame_info) {
    __shared__ uint32_t flags[2 * WORDS * RC_T];  // [pol][word][lane]
    __shared__ uint8_t valid_sh[RC_MAXN];
    const uint32_t f = blockIdx.x, lane = threadIdx.x;
    const uint32_t s = frame_window[f], n = prm.rows * prm.cols;
    const uint32_t base[2] = {seg_off[2 * s], seg_off[2 * s + 1]}, cnt[2] = {seg_cnt[2 * s], seg_cnt[2 * s + 1]};
    const uint32_t nk[2] = {win_info[4 * (size_t) s + 1], win_info[4 * (size_t) s + 2]};
    const bool unsupported = ECAL_WIN_STATUS(win_info[4 * (size_t) s + 3]) == 4 || nk[0] > RC_MAXK || nk[1] > RC_MAXK;
    {   // (the other launch's frame?)
        const bool small_fits = nk[0] <= 32u * RC_WORDS_SMALL && nk[1] <= 32u * RC_WORDS_SMALL;
        if ((WORDS == RC_WORDS_SMALL) != small_fits) return;
    }
    const uint32_t words_used = unsupported ? 0u : ((nk[0] > nk[1] ? nk[0] : nk[1]) + 31u) / 32u;   // (labels of kept clusters are < nk)
    double R[9], t[3];
#pragma unroll
    for (int i = 0; i < 9; i++) R[i] = pose[12 * (size_t) f + i];
#pragma unroll
    for (int i = 0; i < 3; i++) t[i] = pose[12 * (size_t) f + 9 + i];
    const double inlier = 3;

    for (uint32_t k0 = 0; k0 < n; k0 += RC_T) {
        const uint32_t k = k0 + lane;
        bool ok = k < n && !unsupported;
        double cx = 0, cy = 0, radius[4] = {0, 0, 0, 0}, search = 0;
        if (ok) {
            const double *c = landmarks + 3 * k;
            const double skew = prm.circle_radius / __dsqrt_rn(2.0);
            double u[5], v[5];
            project_point(R, t, prm, (float) c[0], (float) c[1], (float) c[2], &u[0], &v[0]);
            project_point(R, t, prm, (float) (c[0] + skew), (float) (c[1] + skew), (float) c[2], &u[1], &v[1]);
            project_point(R, t, prm, (float) (c[0] + skew), (float) (c[1] - skew), (float) c[2], &u[2], &v[2]);
            project_point(R, t, prm, (float) (c[0] - skew), (float) (c[1] - skew), (float) c[2], &u[3], &v[3]);
            project_point(R, t, prm, (float) (c[0] - skew), (float) (c[1] + skew), (float) c[2], &u[4], &v[4]);
            cx = u[0];
            cy = v[0];
            if (cx >= prm.width || cy >= prm.height || cx < 0 || cy < 0) ok = false;  // :457-461
            double max_radius = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const double dx = u[i + 1] - cx, dy = v[i + 1] - cy;
                radius[i] = __dsqrt_rn(dx * dx + dy * dy);
                if (radius[i] > max_radius) max_radius = radius[i];
            }
            search = (max_radius + inlier) * (max_radius + inlier);
        }
        for (uint32_t pw = 0; pw < 2; pw++)
            for (uint32_t w = 0; w < words_used; w++) flags[(pw * WORDS + w) * RC_T + lane] = 0;
        // inliers -> clusters.  The trip counts are wave-uniform; a lane without a circle just does not mark.
        for (int pol = 0; pol < 2; pol++) {
            const double2 *pts = xy + base[pol];
            const int32_t *lab = kept_labels + base[pol];
            for (uint32_t i0 = 0; i0 < cnt[pol]; i0 += RC_T) {
              const uint32_t mine = i0 + lane < cnt[pol] ? i0 + lane : cnt[pol] - 1u;
              const double2 e_my = pts[mine];
              const int32_t l_my = lab[mine];
              const uint32_t nj = cnt[pol] - i0 < (uint32_t) RC_T ? cnt[pol] - i0 : (uint32_t) RC_T;
              for (uint32_t j = 0; j < nj; j++) {
                double2 e;
                e.x = rc_lane_f64(e_my.x, j);
                e.y = rc_lane_f64(e_my.y, j);
                const int32_t l = __builtin_amdgcn_readlane(l_my, (int) j);
                const double dx = e.x - cx, dy = e.y - cy;
                const double d2 = dx * dx + dy * dy;
                if (!ok || l < 0 || !(d2 < search)) continue;
                const double distance = __dsqrt_rn(d2);
                int idx = 0;
                if (dx >= 0 && dy >= 0) idx = 0;
                else if (dx >= 0 && dy <= 0) idx = 1;
                else if (dx <= 0 && dy <= 0) idx = 2;
                else if (dx <= 0 && dy >= 0) idx = 3;
                const double rq = idx == 0 ? radius[0] : idx == 1 ? radius[1] : idx == 2 ? radius[2] : radius[3];
                if (fabs(distance - rq) <= inlier)
                    flags[((uint32_t) pol * WORDS + ((uint32_t) l >> 5)) * RC_T + lane] |= 1u << (l & 31);
              }
            }
        }
        // whole clusters -> the nine sums of fitCircle
        double sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0, sxxx = 0, syyy = 0, sxyy = 0, sxxy = 0;
        uint32_t members[2] = {0, 0};
        for (int pol = 0; pol < 2; pol++) {
            const double2 *pts = xy + base[pol];
            const int32_t *lab = kept_labels + base[pol];
            uint32_t m = 0;
            for (uint32_t i0 = 0; i0 < cnt[pol]; i0 += RC_T) {
              const uint32_t mine = i0 + lane < cnt[pol] ? i0 + lane : cnt[pol] - 1u;
              const double2 e_my = pts[mine];
              const int32_t l_my = lab[mine];
              const uint32_t nj = cnt[pol] - i0 < (uint32_t) RC_T ? cnt[pol] - i0 : (uint32_t) RC_T;
              for (uint32_t j = 0; j < nj; j++) {
                double2 e;
                e.x = rc_lane_f64(e_my.x, j);
                e.y = rc_lane_f64(e_my.y, j);
                const int32_t l = __builtin_amdgcn_readlane(l_my, (int) j);
                if (!ok || l < 0) continue;
                if (!((flags[((uint32_t) pol * WORDS + ((uint32_t) l >> 5)) * RC_T + lane] >> (l & 31)) & 1u)) continue;
                m++;
                sx += e.x;
                sy += e.y;
                const double xx = e.x * e.x, yy = e.y * e.y, xyv = e.x * e.y;
                sxx += xx;
                syy += yy;
                sxy += xyv;
                sxxx += xx * e.x;
                syyy += yy * e.y;
                sxyy += xyv * e.y;
                sxxy += e.x * xyv;
              }
            }
            members[pol] = m;
        }
        if (members[0] < 5 || members[1] < 5) ok = false;  // :560-563
        double out[3] = {__longlong_as_double(0x7ff8000000000000ll), __longlong_as_double(0x7ff8000000000000ll),
                         __longlong_as_double(0x7ff8000000000000ll)};
        if (ok) {
            double A[3][4] = {{2 * sx, 2 * sy, (double) (members[0] + members[1]), sxx + syy},
                              {2 * sxx, 2 * sxy, sx, sxxx + sxyy},
                              {2 * sxy, 2 * syy, sy, sxxy + syyy}};
            double x[3];
            solve3(A, x);
            const double r = __dsqrt_rn(x[0] * x[0] + x[1] * x[1] + x[2]);
            // std::nth_element(radius, radius + 2, radius + 4): the third smallest of the four
            double a = radius[0], b = radius[1], c2 = radius[2], d = radius[3], tmp;
            if (a > b) tmp = a, a = b, b = tmp;
            if (c2 > d) tmp = c2, c2 = d, d = tmp;
            if (a > c2) tmp = a, a = c2, c2 = tmp;   // a = min
            if (b > d) tmp = b, b = d, d = tmp;      // d = max
            const double third = b > c2 ? b : c2;
            const double ex = x[0] - cx, ey = x[1] - cy;
            if (__dsqrt_rn(ex * ex + ey * ey) > 2 * inlier || fabs(r - third) > 1.5 * inlier) {
                ok = false;  // :572-576
            } else {
                out[0] = x[0];
                out[1] = x[1];
                out[2] = r;
            }
        }
        if (k < n) {
            double *o = feat_xyr + 3 * ((size_t) f * n + k);
            o[0] = out[0];
            o[1] = out[1];
            o[2] = out[2];
            feat_valid[(size_t) f * n + k] = ok ? 1u : 0u;
            valid_sh[k] = ok ? 1 : 0;
        }
    }
    __syncthreads();
    if (lane == 0) {
        // border score and the 20 % rule (:587-627)
        const int cols = (int) prm.cols, rows = (int) prm.rows, total = (int) n;
        const int step = (prm.asymmetric ? 2 : 1) * cols;
        int score[4] = {0, 0, 0, 0}, size[4] = {0, 0, 0, 0}, erased = 0;
        for (int i = 0; i < total; i++) erased += valid_sh[i] ? 0 : 1;
        for (int i = 0; i < cols; i++) size[0]++, score[0] += valid_sh[i] ? 0 : 1;
        for (int i = (rows - 1) * cols; i < total; i++) size[1]++, score[1] += valid_sh[i] ? 0 : 1;
        for (int i = 0; i < total; i += step) size[2]++, score[2] += valid_sh[i] ? 0 : 1;
        for (int i = prm.asymmetric ? 2 * cols - 1 : cols - 1; i < total; i += step)
            size[3]++, score[3] += valid_sh[i] ? 0 : 1;
        uint32_t good = 1;
        if (!prm.fit_circle)
            for (int e = 0; e < 4; e++)
                if (score[e] >= size[e] - 1) good = 0;
        if ((double) erased >= 0.2 * (double) (cols * rows)) good = 0;
        frame_info[2 * (size_t) f] = good;
        frame_info[2 * (size_t) f + 1] = (uint32_t) erased;
    }
}

}  // namespace ecal

using namespace ecal;

extern "C" int ecal_rectify_batch_dev(ecal_ctx *ctx, const double *d_xy, const uint32_t *d_seg_off,
                                      const uint32_t *d_seg_cnt, const int32_t *d_kept_labels,
                                      const uint32_t *d_win_info, const uint32_t *d_frame_window, const double *d_pose,
                                      uint32_t F, const double *d_landmarks, const ecal_rectify_params *prm,
                                      double *d_feat_xyr, uint32_t *d_feat_valid, uint32_t *d_frame_info, void *stream) {
    if (!ctx) return ECAL_ERR_INVALID;
    if (F == 0) return ECAL_OK;
    if (!d_xy || !d_seg_off || !d_seg_cnt || !d_kept_labels || !d_win_info || !d_frame_window || !d_pose ||
        !d_landmarks || !prm || !d_feat_xyr || !d_feat_valid || !d_frame_info) {
        ctx->last_error = "null pointer";
        return ECAL_ERR_INVALID;
    }
    if (prm->rows * prm->cols == 0 || prm->rows * prm->cols > RC_MAXN) {
        ctx->last_error = "rows * cols must be in [1, 128]";
        return ECAL_ERR_INVALID;
    }
    RectifyConst c;
    c.fx = prm->fx, c.fy = prm->fy, c.cx = prm->cx, c.cy = prm->cy;
    for (int i = 0; i < 5; i++) c.k[i] = prm->dist[i];
    c.width = prm->width, c.height = prm->height, c.circle_radius = prm->circle_radius;
    c.rows = prm->rows, c.cols = prm->cols, c.asymmetric = prm->asymmetric, c.fit_circle = prm->fit_circle;
    c.model = prm->model == 1 ? 1 : 0;
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(rectify_kernel<RC_WORDS_SMALL>, dim3(F), dim3(RC_T), 0, (hipStream_t) stream, (const double2 *) d_xy, d_seg_off,
                       d_seg_cnt, d_kept_labels, d_win_info, d_frame_window, d_pose, d_landmarks, c, d_feat_xyr,
                       d_feat_valid, d_frame_info);
    hipLaunchKernelGGL(rectify_kernel<RC_WORDS>, dim3(F), dim3(RC_T), 0, (hipStream_t) stream, (const double2 *) d_xy, d_seg_off,
                       d_seg_cnt, d_kept_labels, d_win_info, d_frame_window, d_pose, d_landmarks, c, d_feat_xyr,
                       d_feat_valid, d_frame_info);
    ECAL_HIP_TRY(ctx, hipGetLastError());
    return ECAL_OK;
}
