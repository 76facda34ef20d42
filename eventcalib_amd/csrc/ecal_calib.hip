// Init calibration on calibration views: per-view reprojection residual / Jacobian blocks, the Schur-reduced
// Levenberg-Marquardt solve around them, planar pose initialisation (homography -> IPPE -> LM refinement).
//
// Replaces, for the keyframes the reference selects (event_camera_calib/src/EventCalibIni.cpp:159-183):
//   cv::calibrateCamera(objectPoints, imagePoints, imageSize, K, dist, rvecs, tvecs, flag | CALIB_USE_LU)   :198-199
//   cv::fisheye::calibrate(objectPoints, imagePoints, imageSize, K, dist, rvecs, tvecs, flag)              :188
//   cv::solvePnPRansac(objectPoints[0], imageP, K, dist, rvec, tvec, false, 50, 4.0, 0.99, inliers, IPPE)  :258-259
// OpenCV is third party (>= 4.0, not vendored, not pinned): the algorithms are restated from their published
// form (oracle/calib_oracle.py lists them) — parity with OpenCV itself is unpinned; the kernels are checked
// against that oracle and against synthetic ground truth.
//
// MI355X design.  A view is 36 circles x 2 residuals with 12 + 6 unknown columns: one wave per view, one lane per
// circle.  The lane writes its two rows [J | r] (19 doubles) to LDS and the wave forms the 19 x 19 Gram matrix
// (190 dot products over <= 256 rows, three per lane, sequential over rows -> bit-reproducible).  The normal
// matrix of the whole problem is an arrow: a dense 12 x 12 head for the shared intrinsics and one independent
// 6 x 6 block per view.  OpenCV factorises it densely ((12 + 6V)^2 doubles, 1.2 MB for 64 views); here every
// view eliminates its own 6 x 6 block in its wave (Schur complement) so that what leaves the GPU — and what
// ranks sum with the RCCL all-reduce when views are sharded one batch per GPU — is 170 doubles per evaluation.
// The 12 x 12 reduced system is solved on the host (LU with partial pivoting, as CALIB_USE_LU) and the
// per-view back-substitution runs again one wave per view.
#include <float.h>
#include <math.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <utility>
#include "ecal_ctx.hpp"
#include "calib_models.hpp"

#pragma clang fp contract(off)

namespace ecal {

constexpr int CBK_T = 64;
constexpr uint32_t CB_MAXPTS = 128;
// per-view block record (doubles)
constexpr int CBO_HII = 0, CBO_HIV = 144, CBO_HVV = 216, CBO_GI = 252, CBO_GV = 264, CBO_COST = 270, CB_BLOCK = 272;
// reduced record: S 144 | g 12 | diag(Hii) 12 | cost | points
constexpr int CRO_S = 0, CRO_G = 144, CRO_D = 156, CRO_COST = 168, CRO_NPTS = 169, CB_RED = 170, CB_RED_STRIDE = 176;

struct CalibConst {
    int model;
    uint32_t free_mask;  // bit j = intrinsics slot j is optimised
    int fix_aspect;
    double aspect;
};

struct CalibLds {
    double rows[2 * CB_MAXPTS][CB_NC];
    double G[CB_NC][CB_NC];
    double L[6][6];
    double W[CB_NI][6];
    double vec[32];
    double nx[CB_MAXPTS], ny[CB_MAXPTS];
    uint32_t inl[CB_MAXPTS];
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// G[a][b] = sum_r rows[r][a] * rows[r][b] for a, b < nc (sequential over r: reproducible)
__device__ __forceinline__ void wave_gram(CalibLds &S, int nrows, int nc) {
    const int lane = threadIdx.x;
    const int ne = nc * (nc + 1) / 2;
    for (int e = lane; e < ne; e += CBK_T) {
        int a = 0, k = e;
        while (k >= nc - a) {
            k -= nc - a;
            a++;
        }
        const int b = a + k;
        double s = 0;
        for (int r = 0; r < nrows; r++) s += S.rows[r][a] * S.rows[r][b];
        S.G[a][b] = s;
        S.G[b][a] = s;
    }
    __syncthreads();
}

// in-place Cholesky of the n x n SPD matrix A (row-major, leading dimension ld) — one lane
__device__ __forceinline__ void chol_inplace(double *A, int n, int ld) {
    for (int j = 0; j < n; j++) {
        double d = A[j * ld + j];
        for (int k = 0; k < j; k++) d -= A[j * ld + k] * A[j * ld + k];
        d = sqrt(d);
        A[j * ld + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[i * ld + j];
            for (int k = 0; k < j; k++) s -= A[i * ld + k] * A[j * ld + k];
            A[i * ld + j] = s / d;
        }
    }
}
__device__ __forceinline__ void chol_solve(const double *Lm, int n, int ld, double *x) {
    for (int i = 0; i < n; i++) {
        double s = x[i];
        for (int k = 0; k < i; k++) s -= Lm[i * ld + k] * x[k];
        x[i] = s / Lm[i * ld + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = x[i];
        for (int k = i + 1; k < n; k++) s -= Lm[k * ld + i] * x[k];
        x[i] = s / Lm[i * ld + i];
    }
}

// rows [J | r] of every point of the view into S.rows; returns nothing (cost is taken from the rows)
template <bool JAC>
__device__ __forceinline__ void fill_rows(CalibLds &S, const CalibConst &cc, const double *in, const double *q, const double *obj,
                                          const double *img, uint32_t n, int pose_only) {
    RodriguesTerms T;
    rodrigues_terms(q, T);
    for (uint32_t pt = threadIdx.x; pt < n; pt += CBK_T) {
        const double M[3] = {obj[3 * pt], obj[3 * pt + 1], obj[3 * pt + 2]};
        double Y[3], RM[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            RM[k] = T.R[3 * k] * M[0] + T.R[3 * k + 1] * M[1] + T.R[3 * k + 2] * M[2];
            Y[k] = RM[k] + q[3 + k];
        }
        double u, v, du[CB_NI], dv[CB_NI], P[2][3];
        project_cam<JAC>(cc.model, in, Y, &u, &v, du, dv, P);
        double *ru = S.rows[2 * pt], *rv = S.rows[2 * pt + 1];
        const bool ok = S.inl[pt] != 0;
        if (JAC) {
            if (cc.fix_aspect) {
                du[1] += cc.aspect * du[0];
                dv[1] += cc.aspect * dv[0];
            }
            double dY[3][3];
            drot_point(T, RM, dY);
            if (!pose_only) {
#pragma unroll
                for (int j = 0; j < CB_NI; j++) {
                    const bool fr = ok && ((cc.free_mask >> j) & 1u);
                    ru[j] = fr ? du[j] : 0.0;
                    rv[j] = fr ? dv[j] : 0.0;
                }
            }
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const double a = P[0][0] * dY[i][0] + P[0][1] * dY[i][1] + P[0][2] * dY[i][2];
                const double b = P[1][0] * dY[i][0] + P[1][1] * dY[i][1] + P[1][2] * dY[i][2];
                ru[CB_NI + i] = ok ? a : 0.0;
                rv[CB_NI + i] = ok ? b : 0.0;
                ru[CB_NI + 3 + i] = ok ? P[0][i] : 0.0;
                rv[CB_NI + 3 + i] = ok ? P[1][i] : 0.0;
            }
        }
        ru[CB_NP] = ok ? u - img[2 * pt] : 0.0;
        rv[CB_NP] = ok ? v - img[2 * pt + 1] : 0.0;
    }
    __syncthreads();
}

// sum of squared residuals, sequential over the rows (the same order in every evaluation mode)
__device__ __forceinline__ double rows_cost(CalibLds &S, uint32_t n) {
    if (threadIdx.x == 0) {
        double s = 0;
        for (uint32_t r = 0; r < 2 * n; r++) s += S.rows[r][CB_NP] * S.rows[r][CB_NP];
        S.vec[31] = s;
    }
    __syncthreads();
    const double c = S.vec[31];
    __syncthreads();
    return c;
}

__device__ __forceinline__ void load_intr(const CalibConst &cc, const double *d_intr, double *in) {
#pragma unroll
    for (int j = 0; j < CB_NI; j++) in[j] = d_intr[j];
    if (cc.fix_aspect) in[0] = in[1] * cc.aspect;
}

// ---- per-view blocks ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(CBK_T) void calib_eval_kernel(const double *obj, uint32_t n, const double *img, CalibConst cc,
                                                           const double *d_intr, const double *view_params, int with_jac,
                                                           double *blocks) {
    __shared__ CalibLds S;
    const uint32_t v = blockIdx.x;
    double in[CB_NI], q[6];
    load_intr(cc, d_intr, in);
#pragma unroll
    for (int k = 0; k < 6; k++) q[k] = view_params[6 * v + k];
    for (uint32_t pt = threadIdx.x; pt < n; pt += CBK_T) S.inl[pt] = 1;
    __syncthreads();
    double *B = blocks + (size_t) v * CB_BLOCK;
    if (with_jac) {
        fill_rows<true>(S, cc, in, q, obj, img + (size_t) v * n * 2, n, 0);
        wave_gram(S, 2 * n, CB_NC);
        for (int e = threadIdx.x; e < CB_BLOCK; e += CBK_T) {
            double val = 0;
            if (e < CBO_HIV) val = S.G[e / CB_NI][e % CB_NI];
            else if (e < CBO_HVV) val = S.G[(e - CBO_HIV) / 6][CB_NI + (e - CBO_HIV) % 6];
            else if (e < CBO_GI) val = S.G[CB_NI + (e - CBO_HVV) / 6][CB_NI + (e - CBO_HVV) % 6];
            else if (e < CBO_GV) val = S.G[e - CBO_GI][CB_NP];
            else if (e < CBO_COST) val = S.G[CB_NI + e - CBO_GV][CB_NP];
            else if (e == CBO_COST) continue;
            B[e] = val;
        }
    } else {
        fill_rows<false>(S, cc, in, q, obj, img + (size_t) v * n * 2, n, 0);
    }
    const double c = rows_cost(S, n);
    if (threadIdx.x == 0) B[CBO_COST] = c;
}

// Schur complement of one view's 6 x 6 block with the LM diagonal (1 + lambda) applied to it
__global__ __launch_bounds__(CBK_T) void calib_schur_kernel(const double *blocks, uint32_t n, double lambda, double *view_red) {
    __shared__ double B[CB_BLOCK];
    __shared__ double L[6][6];
    __shared__ double W[CB_NI][6];
    const uint32_t v = blockIdx.x;
    for (int e = threadIdx.x; e < CB_BLOCK; e += CBK_T) B[e] = blocks[(size_t) v * CB_BLOCK + e];
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) L[i][j] = B[CBO_HVV + 6 * i + j] * (i == j ? 1.0 + lambda : 1.0);
        chol_inplace(&L[0][0], 6, 6);
    }
    __syncthreads();
    if (threadIdx.x < CB_NI) {
        double x[6];
        for (int k = 0; k < 6; k++) x[k] = B[CBO_HIV + 6 * threadIdx.x + k];
        chol_solve(&L[0][0], 6, 6, x);
        for (int k = 0; k < 6; k++) W[threadIdx.x][k] = x[k];
    }
    __syncthreads();
    double *R = view_red + (size_t) v * CB_RED_STRIDE;
    for (int e = threadIdx.x; e < CB_RED; e += CBK_T) {
        double val;
        if (e < CRO_G) {
            const int i = e / CB_NI, j = e % CB_NI;
            double s = B[CBO_HII + e];
            for (int k = 0; k < 6; k++) s -= W[i][k] * B[CBO_HIV + 6 * j + k];
            val = s;
        } else if (e < CRO_D) {
            const int i = e - CRO_G;
            double s = B[CBO_GI + i];
            for (int k = 0; k < 6; k++) s -= W[i][k] * B[CBO_GV + k];
            val = s;
        } else if (e < CRO_COST) {
            val = B[CBO_HII + (e - CRO_D) * (CB_NI + 1)];
        } else if (e == CRO_COST) {
            val = B[CBO_COST];
        } else {
            val = (double) n;
        }
        R[e] = val;
    }
}

// cost-only record (same slots, everything else zero) so that both evaluation modes reduce the same buffer
__global__ __launch_bounds__(CBK_T) void calib_costrec_kernel(const double *blocks, uint32_t n, double *view_red) {
    const uint32_t v = blockIdx.x;
    double *R = view_red + (size_t) v * CB_RED_STRIDE;
    for (int e = threadIdx.x; e < CB_RED; e += CBK_T)
        R[e] = e == CRO_COST ? blocks[(size_t) v * CB_BLOCK + CBO_COST] : (e == CRO_NPTS ? (double) n : 0.0);
}

// out[e] = sum over views: four interleaved partial sums per entry (views v = q mod 4, each in view order), combined
// in the fixed order q = 0..3 — deterministic, and a quarter of the dependent-load chain of a single running sum
constexpr int CB_RED_Q = 4;
__global__ __launch_bounds__(CB_RED_STRIDE * CB_RED_Q) void calib_reduce_kernel(const double *view_red, uint32_t V, double *out) {
    __shared__ double part[CB_RED_Q][CB_RED_STRIDE];
    const int e = threadIdx.x % CB_RED_STRIDE, q = threadIdx.x / CB_RED_STRIDE;
    double s = 0;
    if (e < CB_RED)
        for (uint32_t v = q; v < V; v += CB_RED_Q) s += view_red[(size_t) v * CB_RED_STRIDE + e];
    part[q][e] = s;
    __syncthreads();
    if (q == 0 && e < CB_RED) out[e] = ((part[0][e] + part[1][e]) + part[2][e]) + part[3][e];
}

// back-substitution of one view: x_v = (Hvv')^-1 (g_v - Hvi x_i); cand = prev - scale * x_v
__global__ __launch_bounds__(CBK_T) void calib_update_kernel(const double *blocks, double lambda, const double *x_i, double scale,
                                                             const double *prev, double *cand) {
    const uint32_t v = blockIdx.x;
    if (threadIdx.x != 0) return;
    const double *B = blocks + (size_t) v * CB_BLOCK;
    double Lm[36], x[6];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) Lm[6 * i + j] = B[CBO_HVV + 6 * i + j] * (i == j ? 1.0 + lambda : 1.0);
    chol_inplace(Lm, 6, 6);
    for (int k = 0; k < 6; k++) {
        double s = B[CBO_GV + k];
        for (int i = 0; i < CB_NI; i++) s -= B[CBO_HIV + 6 * i + k] * x_i[i];
        x[k] = s;
    }
    chol_solve(Lm, 6, 6, x);
    for (int k = 0; k < 6; k++) cand[6 * v + k] = prev[6 * v + k] - scale * x[k];
}

// ---- planar pose of one view ----------------------------------------------------------------------------------
// least-squares homography dst ~ H src over the inlier points (Hartley-normalised DLT, h22 = 1); src = obj.xy - c
__device__ __forceinline__ bool wave_homography(CalibLds &S, const double *obj, uint32_t n, double cx, double cy, double *H) {
    const int lane = threadIdx.x;
    // normalisation: mean distance to the centroid -> sqrt(2)
    double sd = 0, mx = 0, my = 0, cnt = 0;
    for (uint32_t pt = lane; pt < n; pt += CBK_T)
        if (S.inl[pt]) {
            mx += S.nx[pt];
            my += S.ny[pt];
            cnt += 1;
        }
    mx = wave_sum(mx);
    my = wave_sum(my);
    cnt = wave_sum(cnt);
    if (cnt < 4) return false;
    mx /= cnt;
    my /= cnt;
    double ss = 0;
    for (uint32_t pt = lane; pt < n; pt += CBK_T)
        if (S.inl[pt]) {
            const double ax = obj[3 * pt] - cx, ay = obj[3 * pt + 1] - cy;
            ss += sqrt(ax * ax + ay * ay);
            const double bx = S.nx[pt] - mx, by = S.ny[pt] - my;
            sd += sqrt(bx * bx + by * by);
        }
    ss = wave_sum(ss);
    sd = wave_sum(sd);
    const double ks = 1.4142135623730951 * cnt / ss, kd = 1.4142135623730951 * cnt / sd;
    for (uint32_t pt = lane; pt < n; pt += CBK_T) {
        double *r0 = S.rows[2 * pt], *r1 = S.rows[2 * pt + 1];
        if (S.inl[pt]) {
            const double a0 = (obj[3 * pt] - cx) * ks, a1 = (obj[3 * pt + 1] - cy) * ks;
            const double b0 = (S.nx[pt] - mx) * kd, b1 = (S.ny[pt] - my) * kd;
            r0[0] = a0; r0[1] = a1; r0[2] = 1; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -b0 * a0; r0[7] = -b0 * a1; r0[8] = b0;
            r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = a0; r1[4] = a1; r1[5] = 1; r1[6] = -b1 * a0; r1[7] = -b1 * a1; r1[8] = b1;
        } else {
            for (int k = 0; k < 9; k++) r0[k] = r1[k] = 0;
        }
    }
    __syncthreads();
    wave_gram(S, 2 * n, 9);
    if (lane == 0) {
        double A[64], b[8];
        for (int i = 0; i < 8; i++) {
            for (int j = 0; j < 8; j++) A[8 * i + j] = S.G[i][j];
            b[i] = S.G[i][8];
        }
        chol_inplace(A, 8, 8);
        chol_solve(A, 8, 8, b);
        for (int i = 0; i < 8; i++) S.vec[i] = b[i];
    }
    __syncthreads();
    double h[9];
    for (int i = 0; i < 8; i++) h[i] = S.vec[i];
    h[8] = 1;
    __syncthreads();
    // H = Td^-1 Hn Ts,  Ts = diag(ks, ks, 1),  Td^-1 = [1/kd 0 mx; 0 1/kd my; 0 0 1]
    const double ikd = 1.0 / kd;
    double Hn[9];
    for (int c = 0; c < 3; c++) {
        const double sc = c < 2 ? ks : 1.0;
        Hn[c] = (h[c] * ikd + mx * h[6 + c]) * sc;
        Hn[3 + c] = (h[3 + c] * ikd + my * h[6 + c]) * sc;
        Hn[6 + c] = h[6 + c] * sc;
    }
    const double inv = 1.0 / Hn[8];
    for (int k = 0; k < 9; k++) H[k] = Hn[k] * inv;
    bool ok = true;
    for (int k = 0; k < 9; k++) ok = ok && isfinite(H[k]);
    return ok;
}

// R (row-major, orthonormal) -> Rodrigues vector, the branches of cv::Rodrigues
__device__ __forceinline__ void rotation_to_rvec(const double *R, double *r) {
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = fmin(1.0, fmax(-1.0, c));
    const double th = acos(c);
    double ax[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
    const double s = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]) * 0.5;
    if (s < 1e-5) {
        if (c > 0) {
            r[0] = r[1] = r[2] = 0;
            return;
        }
        double t[3] = {sqrt(fmax((R[0] + 1) * 0.5, 0.0)), sqrt(fmax((R[4] + 1) * 0.5, 0.0)), sqrt(fmax((R[8] + 1) * 0.5, 0.0))};
        if (R[1] < 0) t[1] = -t[1];
        if (R[2] < 0) t[2] = -t[2];
        if (fabs(t[0]) < fabs(t[1]) && fabs(t[0]) < fabs(t[2]) && ((R[5] > 0) != (t[1] * t[2] > 0))) t[2] = -t[2];
        const double nn = th / sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
        r[0] = t[0] * nn; r[1] = t[1] * nn; r[2] = t[2] * nn;
        return;
    }
    const double k = 0.5 * th / s;
    r[0] = ax[0] * k; r[1] = ax[1] * k; r[2] = ax[2] * k;
}

// IPPE (Collins & Bartoli 2014): both poses from the homography of the centred board plane to normalised image
// coordinates, translation by linear least squares over the inliers, best (normalised reprojection error) first.
__device__ __forceinline__ bool wave_ippe(CalibLds &S, const double *obj, uint32_t n, double *q /*rvec, tvec*/) {
    const int lane = threadIdx.x;
    double cx = 0, cy = 0, cnt = 0;
    for (uint32_t pt = lane; pt < n; pt += CBK_T)
        if (S.inl[pt]) {
            cx += obj[3 * pt];
            cy += obj[3 * pt + 1];
            cnt += 1;
        }
    cx = wave_sum(cx);
    cy = wave_sum(cy);
    cnt = wave_sum(cnt);
    if (cnt < 4) return false;
    cx /= cnt;
    cy /= cnt;
    double H[9];
    if (!wave_homography(S, obj, n, cx, cy, H)) return false;
    const double p = H[2], qq = H[5];
    const double J00 = H[0] - H[6] * p, J01 = H[1] - H[7] * p, J10 = H[3] - H[6] * qq, J11 = H[4] - H[7] * qq;
    const double t = sqrt(p * p + qq * qq + 1);
    const double w0 = p / t, w1 = qq / t, w2 = 1.0 / t;
    // Rv = I + [k]x + [k]x^2 / (1 + w2),  k = (-w1, w0, 0)
    const double k0 = -w1, k1 = w0, f = 1.0 / (1 + w2);
    double Rv[9];
    Rv[0] = 1 - f * k1 * k1; Rv[1] = f * k0 * k1;     Rv[2] = k1;
    Rv[3] = f * k0 * k1;     Rv[4] = 1 - f * k0 * k0; Rv[5] = -k0;
    Rv[6] = -k1;             Rv[7] = k0;              Rv[8] = 1 - f * (k0 * k0 + k1 * k1);
    // B = [I2 | -v] Rv (third column vanishes)
    const double B00 = Rv[0] - p * Rv[6], B01 = Rv[1] - p * Rv[7], B10 = Rv[3] - qq * Rv[6], B11 = Rv[4] - qq * Rv[7];
    const double idet = 1.0 / (B00 * B11 - B01 * B10);
    const double A00 = (B11 * J00 - B01 * J10) * idet, A01 = (B11 * J01 - B01 * J11) * idet;
    const double A10 = (-B10 * J00 + B00 * J10) * idet, A11 = (-B10 * J01 + B00 * J11) * idet;
    const double m00 = A00 * A00 + A10 * A10, m01 = A00 * A01 + A10 * A11, m11 = A01 * A01 + A11 * A11;
    const double Tr = m00 + m11, Dt = m00 * m11 - m01 * m01;
    const double gamma = sqrt(0.5 * (Tr + sqrt(fmax(Tr * Tr - 4 * Dt, 0.0))));
    const double ig = 1.0 / gamma;
    const double r00 = A00 * ig, r01 = A01 * ig, r10 = A10 * ig, r11 = A11 * ig;
    const double b1 = sqrt(fmax(0.0, 1 - r00 * r00 - r10 * r10));
    double b2 = sqrt(fmax(0.0, 1 - r01 * r01 - r11 * r11));
    if (r00 * r01 + r10 * r11 > 0) b2 = -b2;
    double best_err = 0, bestq[6];
    bool have = false;
    for (int sol = 0; sol < 2; sol++) {
        const double sg = sol == 0 ? 1.0 : -1.0;
        const double c1[3] = {r00, r10, sg * b1}, c2[3] = {r01, r11, sg * b2};
        double c3[3];
        cross3(c1, c2, c3);
        double R[9];
        for (int i = 0; i < 3; i++) {
            R[3 * i] = Rv[3 * i] * c1[0] + Rv[3 * i + 1] * c1[1] + Rv[3 * i + 2] * c1[2];
            R[3 * i + 1] = Rv[3 * i] * c2[0] + Rv[3 * i + 1] * c2[1] + Rv[3 * i + 2] * c2[2];
            R[3 * i + 2] = Rv[3 * i] * c3[0] + Rv[3 * i + 1] * c3[1] + Rv[3 * i + 2] * c3[2];
        }
        // translation: [1 0 -u; 0 1 -v] t = [u P2 - P0; v P2 - P1]
        for (uint32_t pt = lane; pt < n; pt += CBK_T) {
            double *r0 = S.rows[2 * pt], *r1 = S.rows[2 * pt + 1];
            if (S.inl[pt]) {
                const double X = obj[3 * pt] - cx, Y = obj[3 * pt + 1] - cy;
                const double P0 = R[0] * X + R[1] * Y, P1 = R[3] * X + R[4] * Y, P2 = R[6] * X + R[7] * Y;
                const double u = S.nx[pt], v = S.ny[pt];
                r0[0] = 1; r0[1] = 0; r0[2] = -u; r0[3] = u * P2 - P0;
                r1[0] = 0; r1[1] = 1; r1[2] = -v; r1[3] = v * P2 - P1;
            } else {
                for (int k = 0; k < 4; k++) r0[k] = r1[k] = 0;
            }
        }
        __syncthreads();
        wave_gram(S, 2 * n, 4);
        double A3[9], tv[3];
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) A3[3 * i + j] = S.G[i][j];
            tv[i] = S.G[i][3];
        }
        __syncthreads();
        chol_inplace(A3, 3, 3);
        chol_solve(A3, 3, 3, tv);
        double err = 0;
        for (uint32_t pt = lane; pt < n; pt += CBK_T)
            if (S.inl[pt]) {
                const double X = obj[3 * pt] - cx, Y = obj[3 * pt + 1] - cy;
                const double P0 = R[0] * X + R[1] * Y + tv[0], P1 = R[3] * X + R[4] * Y + tv[1], P2 = R[6] * X + R[7] * Y + tv[2];
                const double ex = P0 / P2 - S.nx[pt], ey = P1 / P2 - S.ny[pt];
                err += ex * ex + ey * ey;
            }
        err = wave_sum(err);
        if (!isfinite(err)) continue;
        if (!have || err < best_err) {
            have = true;
            best_err = err;
            rotation_to_rvec(R, bestq);
            // undo the centring: t = t_c - R (cx, cy, 0)
            bestq[3] = tv[0] - (R[0] * cx + R[1] * cy);
            bestq[4] = tv[1] - (R[3] * cx + R[4] * cy);
            bestq[5] = tv[2] - (R[6] * cx + R[7] * cy);
        }
    }
    if (!have) return false;
    for (int k = 0; k < 6; k++) q[k] = bestq[k];
    return true;
}

// one evaluation of the pose-only problem: Gram of [J_pose | r] (7 x 7 at columns 12..18) or just the cost
__device__ __forceinline__ double pose_eval(CalibLds &S, const CalibConst &cc, const double *in, const double *q, const double *obj,
                                            const double *img, uint32_t n, bool jac) {
    if (jac) {
        fill_rows<true>(S, cc, in, q, obj, img, n, 1);
        // compact the 7 pose columns to the front so that the Gram runs over 7 columns
        for (uint32_t r = threadIdx.x; r < 2 * n; r += CBK_T)
            for (int k = 0; k < 7; k++) S.rows[r][k] = S.rows[r][CB_NI + k];
        __syncthreads();
        wave_gram(S, 2 * n, 7);
        for (uint32_t r = threadIdx.x; r < 2 * n; r += CBK_T) S.rows[r][CB_NP] = S.rows[r][6];
        __syncthreads();
    } else {
        fill_rows<false>(S, cc, in, q, obj, img, n, 1);
    }
    return rows_cost(S, n);
}

// CvLevMarq on the six pose parameters (cvFindExtrinsicCameraParams2: max 20 iterations, FLT_EPSILON)
__device__ __forceinline__ double wave_refine_pose(CalibLds &S, const CalibConst &cc, const double *in, double *q, const double *obj,
                                                   const double *img, uint32_t n, int max_iter, double eps) {
    int lam_lg10 = -3, it = 0;
    double err = pose_eval(S, cc, in, q, obj, img, n, true);
    if (max_iter <= 0) return err;
    for (;;) {
        double prev[6], JtJ[36], g[6];
        const double prev_err = err;
        for (int k = 0; k < 6; k++) prev[k] = q[k];
        for (int i = 0; i < 6; i++) {
            for (int j = 0; j < 6; j++) JtJ[6 * i + j] = S.G[i][j];
            g[i] = S.G[i][6];
        }
        __syncthreads();
        for (;;) {
            const double lam = pow(10.0, (double) lam_lg10);
            double A[36], x[6];
            for (int i = 0; i < 36; i++) A[i] = JtJ[i];
            for (int i = 0; i < 6; i++) {
                A[7 * i] *= 1 + lam;
                x[i] = g[i];
            }
            chol_inplace(A, 6, 6);
            chol_solve(A, 6, 6, x);
            for (int k = 0; k < 6; k++) q[k] = prev[k] - x[k];
            err = pose_eval(S, cc, in, q, obj, img, n, false);
            if (!(err <= prev_err)) {  // also taken for NaN
                if (++lam_lg10 <= 16) continue;
            }
            break;
        }
        lam_lg10 = max(lam_lg10 - 1, -16);
        it++;
        double dn = 0, pn = 0;
        for (int k = 0; k < 6; k++) {
            dn += (q[k] - prev[k]) * (q[k] - prev[k]);
            pn += prev[k] * prev[k];
        }
        if (it >= max_iter || sqrt(dn) < eps * sqrt(pn)) return err;
        err = pose_eval(S, cc, in, q, obj, img, n, true);
    }
}

// d_valid: [F][n] 0/1 or NULL.  rounds >= 1 consensus rounds with threshold thresh (px); refine_iters LM iterations.
__global__ __launch_bounds__(CBK_T) void view_pose_kernel(const double *obj, uint32_t n, const double *img, const uint32_t *d_valid,
                                                          CalibConst cc, const double *d_intr, double thresh, int rounds,
                                                          int refine_iters, double refine_eps, double *view_params,
                                                          uint32_t *d_inlier, double *d_err, uint32_t *d_ok) {
    __shared__ CalibLds S;
    const uint32_t v = blockIdx.x;
    const double *im = img + (size_t) v * n * 2;
    double in[CB_NI], q[6] = {0, 0, 0, 0, 0, 0};
    CalibConst pc = cc;
    pc.fix_aspect = 0;  // intrinsics are constants here
    load_intr(cc, d_intr, in);
    for (uint32_t pt = threadIdx.x; pt < n; pt += CBK_T) {
        S.inl[pt] = d_valid ? (d_valid[(size_t) v * n + pt] != 0) : 1u;
        undistort_normalized(cc.model, in, im[2 * pt], im[2 * pt + 1], &S.nx[pt], &S.ny[pt]);
    }
    __syncthreads();
    bool ok = false;
    for (int rd = 0; rd < rounds; rd++) {
        ok = wave_ippe(S, obj, n, q);
        if (!ok || !(thresh > 0)) break;
        // inliers of this pose: reprojection error <= thresh px
        uint32_t changed = 0;
        RodriguesTerms T;
        rodrigues_terms(q, T);
        for (uint32_t pt = threadIdx.x; pt < n; pt += CBK_T) {
            double Y[3], u, w, du[1], dv[1], P[2][3];
            for (int k = 0; k < 3; k++) Y[k] = T.R[3 * k] * obj[3 * pt] + T.R[3 * k + 1] * obj[3 * pt + 1] + T.R[3 * k + 2] * obj[3 * pt + 2] + q[3 + k];
            project_cam<false>(cc.model, in, Y, &u, &w, du, dv, P);
            const double ex = u - im[2 * pt], ey = w - im[2 * pt + 1];
            const uint32_t was = S.inl[pt];
            const uint32_t base = d_valid ? (d_valid[(size_t) v * n + pt] != 0) : 1u;
            const uint32_t now = base && (sqrt(ex * ex + ey * ey) <= thresh);
            changed |= (was != now);
            S.inl[pt] = now;
        }
        __syncthreads();
        changed = __any(changed) ? 1u : 0u;
        if (!changed) break;
    }
    double err = 0;
    if (ok && refine_iters > 0) err = wave_refine_pose(S, pc, in, q, obj, im, n, refine_iters, refine_eps);
    else if (ok) {
        fill_rows<false>(S, pc, in, q, obj, im, n, 1);
        err = rows_cost(S, n);
    }
    if (threadIdx.x == 0) {
        for (int k = 0; k < 6; k++) view_params[6 * v + k] = ok ? q[k] : 0.0;
        if (d_err) d_err[v] = err;
        if (d_ok) d_ok[v] = ok ? 1u : 0u;
    }
    if (d_inlier)
        for (uint32_t pt = threadIdx.x; pt < n; pt += CBK_T) d_inlier[(size_t) v * n + pt] = ok ? S.inl[pt] : 0u;
}

// homography of the raw board coordinates to pixels, per view (input of the focal-length initialisation)
__global__ __launch_bounds__(CBK_T) void view_homography_kernel(const double *obj, uint32_t n, const double *img, double *d_H,
                                                                uint32_t *d_ok) {
    __shared__ CalibLds S;
    const uint32_t v = blockIdx.x;
    const double *im = img + (size_t) v * n * 2;
    double cx = 0, cy = 0;
    for (uint32_t pt = threadIdx.x; pt < n; pt += CBK_T) {
        S.inl[pt] = 1;
        S.nx[pt] = im[2 * pt];
        S.ny[pt] = im[2 * pt + 1];
        cx += obj[3 * pt];
        cy += obj[3 * pt + 1];
    }
    __syncthreads();
    cx = wave_sum(cx) / n;
    cy = wave_sum(cy) / n;
    double H[9];
    const bool ok = wave_homography(S, obj, n, cx, cy, H);
    if (threadIdx.x == 0) {
        // H maps (X - c); compose with the translation: H_raw = H * [1 0 -cx; 0 1 -cy; 0 0 1]
        for (int r = 0; r < 3; r++) {
            const double a = H[3 * r], b = H[3 * r + 1], c = H[3 * r + 2];
            d_H[9 * v + 3 * r] = a;
            d_H[9 * v + 3 * r + 1] = b;
            d_H[9 * v + 3 * r + 2] = c - a * cx - b * cy;
        }
        d_ok[v] = ok ? 1u : 0u;
    }
}

}  // namespace ecal

using namespace ecal;

// ================================================================================================================
// host side
// ================================================================================================================
namespace {

uint32_t free_mask_of(int model, uint32_t flags) {
    uint32_t m = 0;
    if (model == 0) {
        m = 0xFFFu;
        if (flags & ECAL_CALIB_FIX_ASPECT_RATIO) m &= ~1u;
        if (flags & ECAL_CALIB_FIX_PRINCIPAL_POINT) m &= ~((1u << 2) | (1u << 3));
        if (flags & ECAL_CALIB_ZERO_TANGENT_DIST) m &= ~((1u << 6) | (1u << 7));
        if (flags & ECAL_CALIB_FIX_K1) m &= ~(1u << 4);
        if (flags & ECAL_CALIB_FIX_K2) m &= ~(1u << 5);
        if (flags & ECAL_CALIB_FIX_K3) m &= ~(1u << 8);
        if (flags & ECAL_CALIB_FIX_K4) m &= ~(1u << 9);
        if (flags & ECAL_CALIB_FIX_K5) m &= ~(1u << 10);
        if (flags & ECAL_CALIB_FIX_K6) m &= ~(1u << 11);
    } else {
        m = 0x1FFu;
        if (flags & ECAL_CALIB_FIX_PRINCIPAL_POINT) m &= ~((1u << 2) | (1u << 3));
        if (flags & ECAL_CALIB_FIX_SKEW) m &= ~(1u << 4);
        if (flags & ECAL_CALIB_FIX_K1) m &= ~(1u << 5);
        if (flags & ECAL_CALIB_FIX_K2) m &= ~(1u << 6);
        if (flags & ECAL_CALIB_FIX_K3) m &= ~(1u << 7);
        if (flags & ECAL_CALIB_FIX_K4) m &= ~(1u << 8);
    }
    return m;
}

CalibConst make_const(int model, uint32_t flags, double aspect) {
    CalibConst cc;
    cc.model = model;
    cc.free_mask = free_mask_of(model, flags);
    cc.fix_aspect = (model == 0 && (flags & ECAL_CALIB_FIX_ASPECT_RATIO)) ? 1 : 0;
    cc.aspect = aspect;
    return cc;
}

// dense LU with partial pivoting (what CALIB_USE_LU selects in CvLevMarq::step); false if singular
bool lu_solve(double *A, double *b, int n) {
    for (int c = 0; c < n; c++) {
        int piv = c;
        for (int r = c + 1; r < n; r++)
            if (fabs(A[r * n + c]) > fabs(A[piv * n + c])) piv = r;
        if (!(fabs(A[piv * n + c]) > 0)) return false;
        if (piv != c) {
            for (int k = 0; k < n; k++) std::swap(A[c * n + k], A[piv * n + k]);
            std::swap(b[c], b[piv]);
        }
        for (int r = c + 1; r < n; r++) {
            const double f = A[r * n + c] / A[c * n + c];
            for (int k = c; k < n; k++) A[r * n + k] -= f * A[c * n + k];
            b[r] -= f * b[c];
        }
    }
    for (int r = n - 1; r >= 0; r--) {
        double s = b[r];
        for (int k = r + 1; k < n; k++) s -= A[r * n + k] * b[k];
        b[r] = s / A[r * n + r];
    }
    return true;
}

struct CalibWork {
    ecal_ctx *ctx;
    hipStream_t st;
    CalibConst cc;
    uint32_t V, n;
    double *d_obj, *d_img, *d_intr, *d_view[2], *d_blocks[2], *d_view_red, *d_red, *d_xi;
    double *h_red;  // pinned
    // pinned staging of the small uploads (the intrinsics of an evaluation, two slots in turn; the reduced step): a hipMemcpyAsync
    // from the caller's stack goes through the runtime's pageable path — a staging copy and a wait of its own, three to four times
    // per LM iteration, each as long as the kernels it sits between
    double *h_intr[2], *h_xi;
    int intr_slot = 0;
    // the two view-parameter arrays of the step test (accepted state, candidate) ride on the NEXT reduce_to_host's synchronisation
    // when the caller asks for them here (a download + wait of their own was a third of an LM iteration's round trips)
    double *fetch_views[2] = {nullptr, nullptr};
    int fetch_from[2] = {0, 0};
    const ecal_calib_options *opt;
    int jac_evals = 0, err_evals = 0;
};

// (every use of a slot is followed by a synchronisation of the stream before the slot comes round again: cost_of / reduced_step)
static const double *stage_intr(CalibWork &w, const double *intr) {
    double *h = w.h_intr[w.intr_slot ^= 1];
    for (int j = 0; j < CB_NI; j++) h[j] = intr[j];
    return h;
}

int launch_eval(CalibWork &w, const double *intr, int which_view, int which_blocks, int with_jac) {
    ECAL_HIP_TRY(w.ctx, hipMemcpyAsync(w.d_intr, stage_intr(w, intr), CB_NI * sizeof(double), hipMemcpyHostToDevice, w.st));
    if (w.V) hipLaunchKernelGGL(calib_eval_kernel, dim3(w.V), dim3(CBK_T), 0, w.st, w.d_obj, w.n, w.d_img, w.cc, w.d_intr,
                       w.d_view[which_view], with_jac, w.d_blocks[which_blocks]);
    if (with_jac) w.jac_evals++;
    else w.err_evals++;
    return ECAL_OK;
}

// view_red -> red (+ all-reduce over ranks) -> host
int reduce_to_host(CalibWork &w) {
    hipLaunchKernelGGL(calib_reduce_kernel, dim3(1), dim3(CB_RED_STRIDE * CB_RED_Q), 0, w.st, w.d_view_red, w.V, w.d_red);
    if (w.opt->allreduce) {
        if (w.opt->allreduce(w.opt->allreduce_user, w.d_red, CB_RED, w.st) != 0) {
            w.ctx->last_error = "all-reduce hook failed";
            return ECAL_ERR_HIP;
        }
    }
    ECAL_HIP_TRY(w.ctx, hipMemcpyAsync(w.h_red, w.d_red, CB_RED * sizeof(double), hipMemcpyDeviceToHost, w.st));
    if (w.fetch_views[0] && w.V) {
        for (int k = 0; k < 2; k++)
            ECAL_HIP_TRY(w.ctx, hipMemcpyAsync(w.fetch_views[k], w.d_view[w.fetch_from[k]], 6 * (size_t) w.V * sizeof(double), hipMemcpyDeviceToHost, w.st));
    }
    w.fetch_views[0] = w.fetch_views[1] = nullptr;
    ECAL_HIP_TRY(w.ctx, hipStreamSynchronize(w.st));
    return ECAL_OK;
}

int cost_of(CalibWork &w, int which_blocks, double *cost, double *npts) {
    if (w.V) hipLaunchKernelGGL(calib_costrec_kernel, dim3(w.V), dim3(CBK_T), 0, w.st, w.d_blocks[which_blocks], w.n, w.d_view_red);
    int rc = reduce_to_host(w);
    if (rc) return rc;
    *cost = w.h_red[CRO_COST];
    if (npts) *npts = w.h_red[CRO_NPTS];
    return ECAL_OK;
}

// reduced step: x_i (12, zero on fixed slots) from the blocks in which_blocks with LM parameter lambda
int reduced_step(CalibWork &w, int which_blocks, double lambda, double *x_i) {
    if (w.V) hipLaunchKernelGGL(calib_schur_kernel, dim3(w.V), dim3(CBK_T), 0, w.st, w.d_blocks[which_blocks], w.n, lambda, w.d_view_red);
    int rc = reduce_to_host(w);
    if (rc) return rc;
    int idx[CB_NI], m = 0;
    for (int j = 0; j < CB_NI; j++)
        if ((w.cc.free_mask >> j) & 1u) idx[m++] = j;
    double A[CB_NI * CB_NI], b[CB_NI];
    for (int i = 0; i < m; i++) {
        for (int j = 0; j < m; j++) A[i * m + j] = w.h_red[CRO_S + idx[i] * CB_NI + idx[j]];
        A[i * m + i] += lambda * w.h_red[CRO_D + idx[i]];
        b[i] = w.h_red[CRO_G + idx[i]];
    }
    for (int j = 0; j < CB_NI; j++) x_i[j] = 0;
    if (m > 0) {
        if (!lu_solve(A, b, m)) {
            w.ctx->last_error = "calibration: singular reduced system";
            return ECAL_ERR_INVALID;
        }
        for (int i = 0; i < m; i++) {
            if (!std::isfinite(b[i])) {
                w.ctx->last_error = "calibration: non-finite step (degenerate views)";
                return ECAL_ERR_INVALID;
            }
            x_i[idx[i]] = b[i];
        }
    }
    return ECAL_OK;
}

int launch_update(CalibWork &w, int which_blocks, double lambda, const double *x_i, double scale, int from_view, int to_view) {
    for (int j = 0; j < CB_NI; j++) w.h_xi[j] = x_i[j];
    ECAL_HIP_TRY(w.ctx, hipMemcpyAsync(w.d_xi, w.h_xi, CB_NI * sizeof(double), hipMemcpyHostToDevice, w.st));
    if (w.V) hipLaunchKernelGGL(calib_update_kernel, dim3(w.V), dim3(CBK_T), 0, w.st, w.d_blocks[which_blocks], lambda, w.d_xi, scale,
                       w.d_view[from_view], w.d_view[to_view]);
    return ECAL_OK;
}

int launch_pose(CalibWork &w, const double *intr, int which_view, const uint32_t *d_valid, double thresh, int rounds, int refine_iters,
                uint32_t *d_inl, double *d_err, uint32_t *d_ok) {
    ECAL_HIP_TRY(w.ctx, hipMemcpyAsync(w.d_intr, stage_intr(w, intr), CB_NI * sizeof(double), hipMemcpyHostToDevice, w.st));
    if (w.V) hipLaunchKernelGGL(view_pose_kernel, dim3(w.V), dim3(CBK_T), 0, w.st, w.d_obj, w.n, w.d_img, d_valid, w.cc, w.d_intr, thresh,
                       rounds, refine_iters, (double) FLT_EPSILON, w.d_view[which_view], d_inl, d_err, d_ok);
    return ECAL_OK;
}

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

extern "C" int ecal_calib_view_blocks_dev(ecal_ctx *ctx, const double *d_obj, uint32_t n_pts, const double *d_img, uint32_t n_views,
                                          int model, uint32_t flags, double aspect_ratio, const double *d_intr,
                                          const double *d_view_params, int with_jacobian, double *d_blocks, void *stream) {
    if (!ctx) return ECAL_ERR_INVALID;
    if (!d_obj || !d_img || !d_intr || !d_view_params || !d_blocks || n_pts == 0 || n_pts > CB_MAXPTS || (model != 0 && model != 1)) {
        ctx->last_error = "ecal_calib_view_blocks_dev: bad argument";
        return ECAL_ERR_INVALID;
    }
    if (n_views == 0) return ECAL_OK;
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(calib_eval_kernel, dim3(n_views), dim3(CBK_T), 0, (hipStream_t) stream, d_obj, n_pts, d_img,
                       make_const(model, flags, aspect_ratio), d_intr, d_view_params, with_jacobian, d_blocks);
    ECAL_HIP_TRY(ctx, hipGetLastError());
    return ECAL_OK;
}

extern "C" int ecal_pnp_batch_dev(ecal_ctx *ctx, const double *d_obj, uint32_t n_pts, const double *d_img, const uint32_t *d_valid,
                                  uint32_t n_frames, int model, const double *d_intr, double reproj_thresh, int rounds,
                                  int refine_iters, double *d_pose /*[F][6]*/, uint32_t *d_inlier, double *d_err, uint32_t *d_ok,
                                  void *stream) {
    if (!ctx) return ECAL_ERR_INVALID;
    if (!d_obj || !d_img || !d_intr || !d_pose || n_pts < 4 || n_pts > CB_MAXPTS || (model != 0 && model != 1) || rounds < 1) {
        ctx->last_error = "ecal_pnp_batch_dev: bad argument";
        return ECAL_ERR_INVALID;
    }
    if (n_frames == 0) return ECAL_OK;
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(view_pose_kernel, dim3(n_frames), dim3(CBK_T), 0, (hipStream_t) stream, d_obj, n_pts, d_img, d_valid,
                       make_const(model, 0, 0.0), d_intr, reproj_thresh, rounds, refine_iters, (double) FLT_EPSILON, d_pose, d_inlier,
                       d_err, d_ok);
    ECAL_HIP_TRY(ctx, hipGetLastError());
    return ECAL_OK;
}

// host-buffer form of ecal_pnp_batch_dev (what host/event_calib_ini.hpp calls)
extern "C" int ecal_pnp_batch(ecal_ctx *ctx, const double *obj, uint32_t n_pts, const double *img, const uint32_t *valid, uint32_t n_frames,
                              int model, const double *intr, double reproj_thresh, int rounds, int refine_iters, double *pose,
                              uint32_t *inlier, double *err, uint32_t *ok) {
    if (!ctx) return ECAL_ERR_INVALID;
    if (!obj || !img || !intr || !pose || n_pts < 4 || n_pts > CB_MAXPTS) {
        ctx->last_error = "ecal_pnp_batch: bad argument";
        return ECAL_ERR_INVALID;
    }
    if (n_frames == 0) return ECAL_OK;
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t F = n_frames, n = n_pts;
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        size_t o = off;
        off += (bytes + 255) / 256 * 256;
        return o;
    };
    const size_t o_obj = carve(3 * n * 8), o_img = carve(2 * n * F * 8), o_valid = carve(n * F * 4), o_intr = carve(CB_NI * 8),
                 o_pose = carve(6 * F * 8), o_inl = carve(n * F * 4), o_err = carve(F * 8), o_ok = carve(F * 4);
    int rc = ecal_ensure(ctx, ctx->calib_scratch, off);
    if (rc) return rc;
    char *base = (char *) ctx->calib_scratch.ptr;
    hipStream_t st = ctx->stream;
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(base + o_obj, obj, 3 * n * 8, hipMemcpyHostToDevice, st));
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(base + o_img, img, 2 * n * F * 8, hipMemcpyHostToDevice, st));
    if (valid) ECAL_HIP_TRY(ctx, hipMemcpyAsync(base + o_valid, valid, n * F * 4, hipMemcpyHostToDevice, st));
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(base + o_intr, intr, CB_NI * 8, hipMemcpyHostToDevice, st));
    rc = ecal_pnp_batch_dev(ctx, (const double *) (base + o_obj), n_pts, (const double *) (base + o_img),
                            valid ? (const uint32_t *) (base + o_valid) : nullptr, n_frames, model, (const double *) (base + o_intr),
                            reproj_thresh, rounds, refine_iters, (double *) (base + o_pose), (uint32_t *) (base + o_inl),
                            (double *) (base + o_err), (uint32_t *) (base + o_ok), st);
    if (rc) return rc;
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(pose, base + o_pose, 6 * F * 8, hipMemcpyDeviceToHost, st));
    if (inlier) ECAL_HIP_TRY(ctx, hipMemcpyAsync(inlier, base + o_inl, n * F * 4, hipMemcpyDeviceToHost, st));
    if (err) ECAL_HIP_TRY(ctx, hipMemcpyAsync(err, base + o_err, F * 8, hipMemcpyDeviceToHost, st));
    if (ok) ECAL_HIP_TRY(ctx, hipMemcpyAsync(ok, base + o_ok, F * 4, hipMemcpyDeviceToHost, st));
    ECAL_HIP_TRY(ctx, hipStreamSynchronize(st));
    return ECAL_OK;
}

extern "C" void ecal_calib_default_options(ecal_calib_options *o) {
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->model = 0;
    o->flags = 0;
    o->aspect_ratio = 1.0;
    o->max_iter = 0;
    o->eps = 0;
}

extern "C" int ecal_calibrate_views(ecal_ctx *ctx, const double *obj, uint32_t n_pts, const double *img, uint32_t n_views, double width,
                                    double height, const ecal_calib_options *opt_in, ecal_calib_result *res, double *rvecs, double *tvecs,
                                    double *per_view_err) {
    const ecal_range range__(ctx, "ecal_calibrate_views");
    if (!ctx) return ECAL_ERR_INVALID;
    const ecal_calib_options *opt = opt_in;   // opt->allreduce == NULL: rank-local, always (ecal_comm_allreduce is explicit)
    if (!obj || (!img && n_views) || !opt || !res || n_pts < 4 || n_pts > CB_MAXPTS || (opt->model != 0 && opt->model != 1) ||
        !(width > 0) || !(height > 0)) {
        ctx->last_error = "ecal_calibrate_views: bad argument";
        return ECAL_ERR_INVALID;
    }
    for (uint32_t i = 0; i < n_pts; i++)
        if (obj[3 * i + 2] != 0.0) {
            ctx->last_error = "ecal_calibrate_views: the board must lie in z = 0";
            return ECAL_ERR_INVALID;
        }
    if (n_views == 0 && !opt->allreduce) {
        ctx->last_error = "ecal_calibrate_views: no views";
        return ECAL_ERR_INVALID;
    }
    const double t_begin = now_s();
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    CalibWork w;
    w.ctx = ctx;
    w.st = ctx->stream;
    w.V = n_views;
    w.n = n_pts;
    w.opt = opt;
    w.cc = make_const(opt->model, opt->flags, opt->aspect_ratio);
    const uint32_t Va = n_views ? n_views : 1;
    // scratch carve-up (grow-only context buffer)
    size_t off = 0;
    auto carve = [&](size_t doubles) {
        size_t o = off;
        off += (doubles + 15) / 16 * 16;
        return o;
    };
    const size_t o_obj = carve(3 * (size_t) n_pts), o_img = carve(2 * (size_t) n_pts * Va), o_intr = carve(16), o_v0 = carve(6 * (size_t) Va),
                 o_v1 = carve(6 * (size_t) Va), o_b0 = carve((size_t) CB_BLOCK * Va), o_b1 = carve((size_t) CB_BLOCK * Va),
                 o_vr = carve((size_t) CB_RED_STRIDE * Va), o_red = carve(CB_RED_STRIDE), o_xi = carve(16), o_H = carve(9 * (size_t) Va),
                 o_err = carve(Va), o_ok = carve(Va);
    int rc = ecal_ensure(ctx, ctx->calib_scratch, off * sizeof(double));
    if (rc) return rc;
    double *base = (double *) ctx->calib_scratch.ptr;
    w.d_obj = base + o_obj;
    w.d_img = base + o_img;
    w.d_intr = base + o_intr;
    w.d_view[0] = base + o_v0;
    w.d_view[1] = base + o_v1;
    w.d_blocks[0] = base + o_b0;
    w.d_blocks[1] = base + o_b1;
    w.d_view_red = base + o_vr;
    w.d_red = base + o_red;
    w.d_xi = base + o_xi;
    double *d_H = base + o_H, *d_err = base + o_err;
    uint32_t *d_ok = (uint32_t *) (base + o_ok);
    if (!ctx->calib_pinned) ECAL_HIP_TRY(ctx, hipHostMalloc((void **) &ctx->calib_pinned, (CB_RED_STRIDE + 48) * sizeof(double), hipHostMallocDefault));
    w.h_red = ctx->calib_pinned;
    w.h_intr[0] = ctx->calib_pinned + CB_RED_STRIDE;
    w.h_intr[1] = w.h_intr[0] + 16;
    w.h_xi = w.h_intr[1] + 16;
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(w.d_obj, obj, 3 * (size_t) n_pts * sizeof(double), hipMemcpyHostToDevice, w.st));
    if (n_views) ECAL_HIP_TRY(ctx, hipMemcpyAsync(w.d_img, img, 2 * (size_t) n_pts * n_views * sizeof(double), hipMemcpyHostToDevice, w.st));

    double intr[CB_NI];
    for (int j = 0; j < CB_NI; j++) intr[j] = 0;
    std::vector<uint32_t> okv(Va);
    if (opt->flags & ECAL_CALIB_USE_INTRINSIC_GUESS) {
        // cv::CALIB_USE_INTRINSIC_GUESS / cv::fisheye::CALIB_USE_INTRINSIC_GUESS: the caller's res->intr is the start
        for (int j = 0; j < CB_NI; j++) intr[j] = res->intr[j];
        if (!(intr[0] > 0) || !(intr[1] > 0) || !std::isfinite(intr[2]) || !std::isfinite(intr[3])) {
            ctx->last_error = "ecal_calibrate_views: ECAL_CALIB_USE_INTRINSIC_GUESS needs fx, fy > 0 and a principal point in res->intr";
            return ECAL_ERR_INVALID;
        }
        if (w.cc.fix_aspect && opt->model == 0 && opt->aspect_ratio != 0) intr[0] = opt->aspect_ratio * intr[1];
    } else if (opt->model == 0) {
        // cvInitIntrinsicParams2D: principal point at the image centre, focal lengths from the vanishing-point
        // constraints of every view's homography; 2 x 2 normal equations summed over ranks
        const double cx = (width - 1) * 0.5, cy = (height - 1) * 0.5;
        double acc[6] = {0, 0, 0, 0, 0, 0};  // AtA00 AtA01 AtA11 Atb0 Atb1 views
        if (n_views) {
            hipLaunchKernelGGL(view_homography_kernel, dim3(n_views), dim3(CBK_T), 0, w.st, w.d_obj, n_pts, w.d_img, d_H, d_ok);
            std::vector<double> Hh(9 * (size_t) n_views);
            ECAL_HIP_TRY(ctx, hipMemcpyAsync(Hh.data(), d_H, Hh.size() * sizeof(double), hipMemcpyDeviceToHost, w.st));
            ECAL_HIP_TRY(ctx, hipMemcpyAsync(okv.data(), d_ok, n_views * sizeof(uint32_t), hipMemcpyDeviceToHost, w.st));
            ECAL_HIP_TRY(ctx, hipStreamSynchronize(w.st));
            for (uint32_t v = 0; v < n_views; v++) {
                if (!okv[v]) {
                    ctx->last_error = "ecal_calibrate_views: degenerate view (no homography)";
                    return ECAL_ERR_INVALID;
                }
                double H[9];
                for (int k = 0; k < 9; k++) H[k] = Hh[9 * v + k];
                for (int k = 0; k < 3; k++) {
                    H[k] -= H[6 + k] * cx;
                    H[3 + k] -= H[6 + k] * cy;
                }
                double h[3] = {H[0], H[3], H[6]}, vv[3] = {H[1], H[4], H[7]}, d1[3], d2[3];
                for (int k = 0; k < 3; k++) {
                    d1[k] = (h[k] + vv[k]) * 0.5;
                    d2[k] = (h[k] - vv[k]) * 0.5;
                }
                auto nrm = [](double *z) {
                    const double s = 1.0 / sqrt(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]);
                    z[0] *= s; z[1] *= s; z[2] *= s;
                };
                nrm(h); nrm(vv); nrm(d1); nrm(d2);
                const double rowsA[2][2] = {{h[0] * vv[0], h[1] * vv[1]}, {d1[0] * d2[0], d1[1] * d2[1]}};
                const double rb[2] = {-h[2] * vv[2], -d1[2] * d2[2]};
                for (int r = 0; r < 2; r++) {
                    acc[0] += rowsA[r][0] * rowsA[r][0];
                    acc[1] += rowsA[r][0] * rowsA[r][1];
                    acc[2] += rowsA[r][1] * rowsA[r][1];
                    acc[3] += rowsA[r][0] * rb[r];
                    acc[4] += rowsA[r][1] * rb[r];
                }
                acc[5] += 1;
            }
        }
        if (opt->allreduce) {
            ECAL_HIP_TRY(ctx, hipMemcpyAsync(w.d_red, acc, sizeof(acc), hipMemcpyHostToDevice, w.st));
            if (opt->allreduce(opt->allreduce_user, w.d_red, 6, w.st) != 0) {
                ctx->last_error = "all-reduce hook failed";
                return ECAL_ERR_HIP;
            }
            ECAL_HIP_TRY(ctx, hipMemcpyAsync(acc, w.d_red, sizeof(acc), hipMemcpyDeviceToHost, w.st));
            ECAL_HIP_TRY(ctx, hipStreamSynchronize(w.st));
        }
        const double det = acc[0] * acc[2] - acc[1] * acc[1];
        const double f0 = (acc[2] * acc[3] - acc[1] * acc[4]) / det, f1 = (acc[0] * acc[4] - acc[1] * acc[3]) / det;
        double fx = sqrt(fabs(1.0 / f0)), fy = sqrt(fabs(1.0 / f1));
        if (!std::isfinite(fx) || !std::isfinite(fy)) {
            ctx->last_error = "ecal_calibrate_views: focal-length initialisation failed (views too alike)";
            return ECAL_ERR_INVALID;
        }
        if (w.cc.fix_aspect && opt->aspect_ratio != 0) {
            const double tf = (fx + fy) / (opt->aspect_ratio + 1.0);
            fx = opt->aspect_ratio * tf;
            fy = tf;
        }
        intr[0] = fx; intr[1] = fy; intr[2] = cx; intr[3] = cy;
    } else {
        const double f = (width > height ? width : height) / M_PI;
        intr[0] = f; intr[1] = f; intr[2] = width / 2.0 - 0.5; intr[3] = height / 2.0 - 0.5;
    }
    // initial poses: IPPE + LM refinement per view (cvFindExtrinsicCameraParams2 / fisheye CalibrateExtrinsics)
    int cur = 0;  // index of the accepted view parameters / blocks
    if ((rc = launch_pose(w, intr, cur, nullptr, 0.0, 1, 20, nullptr, d_err, d_ok))) return rc;
    if (n_views) {
        ECAL_HIP_TRY(ctx, hipMemcpyAsync(okv.data(), d_ok, n_views * sizeof(uint32_t), hipMemcpyDeviceToHost, w.st));
        ECAL_HIP_TRY(ctx, hipStreamSynchronize(w.st));
        for (uint32_t v = 0; v < n_views; v++)
            if (!okv[v]) {
                ctx->last_error = "ecal_calibrate_views: pose initialisation failed for a view";
                return ECAL_ERR_INVALID;
            }
    }

    const int max_iter = opt->max_iter > 0 ? opt->max_iter : (opt->model == 0 ? 30 : 100);
    const double eps = opt->eps > 0 ? opt->eps : DBL_EPSILON;
    double err = 0, npts = 0, x_i[CB_NI];
    int iters = 0;
    if (opt->model == 0) {
        // CvLevMarq::updateAlt (see oracle/calib_oracle.py::levmarq)
        int lam_lg10 = -3;
        std::vector<double> view_pageable;
        double *view_a = reinterpret_cast<double *>(ecal_fetch_pinned(ctx, 12 * (size_t) Va * sizeof(double)));   // (pinned: see CalibWork::h_intr)
        if (!view_a) {
            view_pageable.resize(12 * (size_t) Va);
            view_a = view_pageable.data();
        }
        double *const view_b = view_a + 6 * (size_t) Va;
        if ((rc = launch_eval(w, intr, cur, cur, 1))) return rc;
        if ((rc = cost_of(w, cur, &err, &npts))) return rc;
        for (;;) {
            const double prev_err = err;
            double cand[CB_NI];
            for (;;) {
                const double lam = pow(10.0, (double) lam_lg10);
                if ((rc = reduced_step(w, cur, lam, x_i))) return rc;
                for (int j = 0; j < CB_NI; j++) cand[j] = intr[j] - x_i[j];
                if (w.cc.fix_aspect) cand[0] = cand[1] * opt->aspect_ratio;
                if ((rc = launch_update(w, cur, lam, x_i, 1.0, cur, cur ^ 1))) return rc;
                if ((rc = launch_eval(w, cand, cur ^ 1, cur ^ 1, 1))) return rc;
                if (view_a) {   // (the step test below wants both states' view parameters: with this cost's download)
                    w.fetch_views[0] = view_a;
                    w.fetch_views[1] = view_b;
                    w.fetch_from[0] = cur;
                    w.fetch_from[1] = cur ^ 1;
                }
                if ((rc = cost_of(w, cur ^ 1, &err, nullptr))) return rc;
                if (!(err <= prev_err)) {
                    if (++lam_lg10 <= 16) continue;
                }
                break;
            }
            lam_lg10 = lam_lg10 - 1 < -16 ? -16 : lam_lg10 - 1;
            iters++;
            // relative parameter change over the whole vector: the view part comes from the device
            double dn = 0, pn = 0;
            {
                const double *const a = view_a, *const b = view_b;   // (arrived with the last cost: the accepted state's and the candidate's)
                if (n_views) {
                    for (size_t k = 0; k < 6 * (size_t) n_views; k++) {
                        dn += (b[k] - a[k]) * (b[k] - a[k]);
                        pn += a[k] * a[k];
                    }
                }
                if (opt->allreduce) {
                    double two[2] = {dn, pn};
                    ECAL_HIP_TRY(ctx, hipMemcpyAsync(w.d_red, two, sizeof(two), hipMemcpyHostToDevice, w.st));
                    if (opt->allreduce(opt->allreduce_user, w.d_red, 2, w.st) != 0) return ECAL_ERR_HIP;
                    ECAL_HIP_TRY(ctx, hipMemcpyAsync(two, w.d_red, sizeof(two), hipMemcpyDeviceToHost, w.st));
                    ECAL_HIP_TRY(ctx, hipStreamSynchronize(w.st));
                    dn = two[0];
                    pn = two[1];
                }
                for (int j = 0; j < CB_NI; j++) {
                    dn += (cand[j] - intr[j]) * (cand[j] - intr[j]);
                    pn += intr[j] * intr[j];
                }
            }
            for (int j = 0; j < CB_NI; j++) intr[j] = cand[j];
            cur ^= 1;  // the candidate (with its blocks) becomes the accepted state
            if (iters >= max_iter || sqrt(dn) < eps * sqrt(pn)) break;
        }
    } else {
        // cv::fisheye::calibrate: Gauss-Newton steps scaled by 1 - (1 - 0.4)^(iter + 1), extrinsics recomputed
        double change = 1;
        while (iters < max_iter && change > eps) {
            if ((rc = launch_eval(w, intr, cur, cur, 1))) return rc;
            if ((rc = reduced_step(w, cur, 0.0, x_i))) return rc;
            const double a2 = 1 - pow(1 - 0.4, iters + 1);
            double cand[CB_NI];
            for (int j = 0; j < CB_NI; j++) cand[j] = intr[j] - a2 * x_i[j];
            double dn = 0, qn = 0;
            for (int j = 0; j < 4; j++) {
                dn += (cand[j] - intr[j]) * (cand[j] - intr[j]);
                qn += cand[j] * cand[j];
            }
            change = sqrt(dn) / sqrt(qn);
            for (int j = 0; j < CB_NI; j++) intr[j] = cand[j];
            if (opt->flags & ECAL_CALIB_RECOMPUTE_EXTRINSIC) {
                if ((rc = launch_pose(w, intr, cur, nullptr, 0.0, 1, 20, nullptr, d_err, d_ok))) return rc;
            } else {
                if ((rc = launch_update(w, cur, 0.0, x_i, a2, cur, cur ^ 1))) return rc;
                cur ^= 1;
            }
            iters++;
        }
        if ((rc = launch_eval(w, intr, cur, cur, 0))) return rc;
        if ((rc = cost_of(w, cur, &err, &npts))) return rc;
    }
    if (opt->model == 0 && npts == 0) {
        double dummy;
        if ((rc = cost_of(w, cur, &dummy, &npts))) return rc;
    }
    // results
    memcpy(res->intr, intr, sizeof(intr));
    res->rms = sqrt(err / npts);
    res->iterations = iters;
    res->jacobian_evaluations = w.jac_evals;
    res->error_evaluations = w.err_evals;
    if (n_views) {
        std::vector<double> q(6 * (size_t) n_views), blk((size_t) CB_BLOCK * n_views);
        ECAL_HIP_TRY(ctx, hipMemcpyAsync(q.data(), w.d_view[cur], q.size() * sizeof(double), hipMemcpyDeviceToHost, w.st));
        ECAL_HIP_TRY(ctx, hipMemcpyAsync(blk.data(), w.d_blocks[cur], blk.size() * sizeof(double), hipMemcpyDeviceToHost, w.st));
        ECAL_HIP_TRY(ctx, hipStreamSynchronize(w.st));
        for (uint32_t v = 0; v < n_views; v++) {
            for (int k = 0; k < 3; k++) {
                if (rvecs) rvecs[3 * v + k] = q[6 * v + k];
                if (tvecs) tvecs[3 * v + k] = q[6 * v + 3 + k];
            }
            if (per_view_err) per_view_err[v] = sqrt(blk[(size_t) v * CB_BLOCK + CBO_COST] / n_pts);
        }
    }
    res->seconds = now_s() - t_begin;
    return ECAL_OK;
}

// cv::fisheye::calibrate's call at EventCalibIni.cpp:186-190, with one start procedure for every front end (the C++ shim
// host/event_calib_ini.hpp and eventcalib_amd/calibrate.py used to differ): FIRST the reference's own start — principal point
// at the image centre, f = max(w, h) / pi, no guess —; only when that fails (the smoothed Gauss-Newton ends on a singular
// system or a non-finite result: a lens far from the 180-degree lens the start assumes, e.g. the example sensor's 55 degrees)
// the radial model is calibrated on the same views with ECAL_CALIB_FISHEYE_PRECALIB_FLAGS and its focal lengths and principal
// point start the fisheye model (CALIB_USE_INTRINSIC_GUESS, as a user of the OpenCV call would).  *start_used = 0 / 1 says
// which.  opt->flags may already carry ECAL_CALIB_USE_INTRINSIC_GUESS (res->intr = the caller's guess): one run, *start_used = 2.
extern "C" int ecal_calibrate_fisheye_views(ecal_ctx *ctx, const double *obj, uint32_t n_pts, const double *img, uint32_t n_views,
                                            double width, double height, const ecal_calib_options *opt, ecal_calib_result *res,
                                            double *rvecs, double *tvecs, double *per_view_err, int *start_used) {
    if (!ctx || !opt || !res) return ECAL_ERR_INVALID;
    if (start_used) *start_used = 2;
    ecal_calib_options fo = *opt;
    fo.model = 1;
    if (fo.flags & ECAL_CALIB_USE_INTRINSIC_GUESS) return ecal_calibrate_views(ctx, obj, n_pts, img, n_views, width, height, &fo, res, rvecs, tvecs, per_view_err);
    auto sane = [](const ecal_calib_result &r) {
        bool ok = std::isfinite(r.rms) && r.intr[0] > 0 && r.intr[1] > 0;
        for (int j = 0; j < 12; j++) ok = ok && std::isfinite(r.intr[j]);
        return ok;
    };
    ecal_calib_result first;
    memset(&first, 0, sizeof(first));
    int rc = ecal_calibrate_views(ctx, obj, n_pts, img, n_views, width, height, &fo, &first, rvecs, tvecs, per_view_err);
    if (rc == ECAL_OK && sane(first)) {
        *res = first;
        if (start_used) *start_used = 0;
        return ECAL_OK;
    }
    if (rc != ECAL_OK && rc != ECAL_ERR_INVALID) return rc;   // (a HIP / communication error is not a failed start)
    ecal_calib_options po = *opt;    // (same all-reduce: every rank takes the same branch, the first run's verdict is global)
    po.model = 0;
    po.flags = ECAL_CALIB_FISHEYE_PRECALIB_FLAGS;
    po.aspect_ratio = 1.0;
    ecal_calib_result pr;
    memset(&pr, 0, sizeof(pr));
    rc = ecal_calibrate_views(ctx, obj, n_pts, img, n_views, width, height, &po, &pr, nullptr, nullptr, nullptr);
    if (rc != ECAL_OK) return rc;
    memset(res, 0, sizeof(*res));
    for (int j = 0; j < 4; j++) res->intr[j] = pr.intr[j];
    fo.flags |= ECAL_CALIB_USE_INTRINSIC_GUESS;
    if (start_used) *start_used = 1;
    return ecal_calibrate_views(ctx, obj, n_pts, img, n_views, width, height, &fo, res, rvecs, tvecs, per_view_err);
}

