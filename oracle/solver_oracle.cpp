// TEST INFRASTRUCTURE ONLY — CPU oracle for the continuous-time calibration solve (see
// dbscan_oracle.cpp for the rules: only tests/, smoke() and bench.py's cpu_baseline may use this).
//
// Restates (reference paths):
//   * event_camera_calib/include/opengv2/event_camera_calib/EventCalibSpline.hpp:36-63 (unDistort)
//     and :158-229 (CalibReprojectionError::operator(), quaternion spline variant, useSO3 = 0), as a
//     template over the scalar type exactly like the reference, differentiated here with a small
//     forward-mode dual number (what ceres::AutoDiffCostFunction<...,1,9,4,4,4,4,3,3,3,3> does,
//     EventCalibSpline.hpp:237-239);
//   * core/spline/include/opengv2/spline/BsplineReal.hpp:107-145 (basis, ders[0]) and :208-231
//     (findSpan); :87-100 knot placement is NOT restated (the spline fit is a "next" row);
//   * core/sensor/src/PinholeCamera.cpp:69-95 inverseRadialDistortion (closed form of
//     Drap & Lefevre), used to initialise k1..k5 at EventCalibSpline.cpp:101-105.
// Third-party arithmetic restated from its published algorithm (source not in the reference tree,
// no version pinned — CMakeLists.txt:45 only says find_package(Ceres)): Ceres 1.x HuberLoss +
// Corrector (rho'' <= 0 branch), EigenQuaternionParameterization (Plus and its 4x3 Jacobian) and the
// Levenberg-Marquardt trust-region loop with Jacobi scaling (trust_region_minimizer.cc,
// levenberg_marquardt_strategy.cc).  Parity of the minimiser is therefore UNPINNED; tests check
// convergence to the same minimum as the product path and to the synthetic ground truth.

#include <cstdint>
#include <cstddef>
#include <cmath>
#include <thread>
#include <vector>
#include <algorithm>

namespace {

constexpr int NP = 37;  // 9 + 4*4 + 4*3 ambient parameters of one residual block

struct Jet {
    double a;
    double v[NP];
    Jet() : a(0) { for (int i = 0; i < NP; i++) v[i] = 0; }
    Jet(double x) : a(x) { for (int i = 0; i < NP; i++) v[i] = 0; }
    static Jet var(double x, int k) {
        Jet j(x);
        j.v[k] = 1;
        return j;
    }
};
inline Jet operator+(const Jet &x, const Jet &y) { Jet r; r.a = x.a + y.a; for (int i = 0; i < NP; i++) r.v[i] = x.v[i] + y.v[i]; return r; }
inline Jet operator-(const Jet &x, const Jet &y) { Jet r; r.a = x.a - y.a; for (int i = 0; i < NP; i++) r.v[i] = x.v[i] - y.v[i]; return r; }
inline Jet operator-(const Jet &x) { Jet r; r.a = -x.a; for (int i = 0; i < NP; i++) r.v[i] = -x.v[i]; return r; }
inline Jet operator*(const Jet &x, const Jet &y) { Jet r; r.a = x.a * y.a; for (int i = 0; i < NP; i++) r.v[i] = x.a * y.v[i] + x.v[i] * y.a; return r; }
inline Jet operator/(const Jet &x, const Jet &y) { Jet r; r.a = x.a / y.a; for (int i = 0; i < NP; i++) r.v[i] = (x.v[i] - r.a * y.v[i]) / y.a; return r; }
inline Jet sqrt(const Jet &x) { Jet r; r.a = std::sqrt(x.a); for (int i = 0; i < NP; i++) r.v[i] = x.v[i] / (2 * r.a); return r; }
inline Jet &operator*=(Jet &x, const Jet &y) { x = x * y; return x; }
inline double sqrt_(double x) { return std::sqrt(x); }
inline Jet sqrt_(const Jet &x) { return sqrt(x); }
inline double sin_(double x) { return std::sin(x); }
inline double cos_(double x) { return std::cos(x); }
inline double atan_(double x) { return std::atan(x); }
inline Jet sin_(const Jet &x) { Jet r; r.a = std::sin(x.a); const double c = std::cos(x.a); for (int i = 0; i < NP; i++) r.v[i] = c * x.v[i]; return r; }
inline Jet cos_(const Jet &x) { Jet r; r.a = std::cos(x.a); const double s = -std::sin(x.a); for (int i = 0; i < NP; i++) r.v[i] = s * x.v[i]; return r; }
inline Jet atan_(const Jet &x) { Jet r; r.a = std::atan(x.a); const double d = 1.0 / (1.0 + x.a * x.a); for (int i = 0; i < NP; i++) r.v[i] = d * x.v[i]; return r; }
inline double tan_(double x) { return std::tan(x); }
inline Jet tan_(const Jet &x) { Jet r; r.a = std::tan(x.a); const double d = 1.0 + r.a * r.a; for (int i = 0; i < NP; i++) r.v[i] = d * x.v[i]; return r; }
inline double val(double x) { return x; }
inline double val(const Jet &x) { return x.a; }

// EventCalibSpline.hpp:158-229 with T = double or Jet.  Quaternion rotation of a vector follows
// Eigen's QuaternionBase::_transformVector (v + w*2(u x v) + u x 2(u x v)).
// The fisheye camera of BASELINE configs[4] — NEW functionality, no reference counterpart (the reference's solver throws for
// anything but the radial model, EventCalibSpline.cpp:97-99): Kannala-Brandt in the inverse form of unDistort — the pixel's
// distorted angle theta_d = r goes to the ray angle theta = r * poly(r^2) (the same five coefficients), and the ray's x, y
// are scaled by tan(theta) / r instead of by poly.  Written as unDistort is, over T; sqrt has no derivative at r = 0 (the
// product's code takes the limit there; the tests stay away from the exact centre).
template <typename T>
T fisheye_scale(const T &r2, const T &poly) {
    const T r = sqrt_(r2);
    return tan_(r * poly) / r;
}

template <typename T>
T residual_functor(const T *intr, const T (*rq)[4], const T (*tp)[3], const double *rb, const double *tb,
                   const double obs[2], const double lm[3], double radius, bool fisheye = false) {
    T q[4], t[3];
    for (int k = 0; k < 4; k++) q[k] = T(rb[0]) * rq[0][k] + T(rb[1]) * rq[1][k] + T(rb[2]) * rq[2][k] + T(rb[3]) * rq[3][k];
    const T nrm = sqrt_(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);  // Qwb_v.normalize()
    for (int k = 0; k < 4; k++) q[k] = q[k] / nrm;
    for (int k = 0; k < 3; k++) t[k] = T(tb[0]) * tp[0][k] + T(tb[1]) * tp[1][k] + T(tb[2]) * tp[2][k] + T(tb[3]) * tp[3][k];
    // unDistort (:36-63)
    T Xc[3];
    Xc[0] = (T(obs[0]) - intr[2]) / intr[0];
    Xc[1] = (T(obs[1]) - intr[3]) / intr[1];
    Xc[2] = T(1.0);
    const T xx = Xc[0] * Xc[0], yy = Xc[1] * Xc[1];
    const T r2 = xx + yy, r4 = r2 * r2, r6 = r4 * r2, r8 = r6 * r2, r10 = r8 * r2;
    T coeff = T(1.0) + intr[4] * r2 + intr[5] * r4 + intr[6] * r6 + intr[7] * r8 + intr[8] * r10;
    if (fisheye) coeff = fisheye_scale(r2, coeff);
    Xc[0] *= coeff;
    Xc[1] *= coeff;
    // :213-224
    const T tx = T(2.0) * q[0], ty = T(2.0) * q[1], tz = T(2.0) * q[2];
    const T twx = tx * q[3], twy = ty * q[3], txx = tx * q[0], txz = tz * q[0], tyy = ty * q[1], tyz = tz * q[1];
    const T r2row[3] = {txz - twy, tyz + twx, T(1.0) - (txx + tyy)};
    const T depth = -t[2] / (r2row[0] * Xc[0] + r2row[1] * Xc[1] + r2row[2] * Xc[2]);
    for (int k = 0; k < 3; k++) Xc[k] *= depth;
    // Xw = Qws * Xc + tws
    T uv[3] = {q[1] * Xc[2] - q[2] * Xc[1], q[2] * Xc[0] - q[0] * Xc[2], q[0] * Xc[1] - q[1] * Xc[0]};
    for (int k = 0; k < 3; k++) uv[k] = uv[k] + uv[k];
    const T c2[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
    T Xw[3];
    for (int k = 0; k < 3; k++) Xw[k] = Xc[k] + q[3] * uv[k] + c2[k] + t[k];
    const T d0 = Xw[0] - T(lm[0]), d1 = Xw[1] - T(lm[1]), d2 = Xw[2] - T(lm[2]);
    return sqrt_(d0 * d0 + d1 * d1 + d2 * d2) - T(radius);
}

// ---- SO3 spline variant (useSO3 = 1): CalibReprojectionError_SO3::operator() (EventCalibSpline.hpp:65-135).
// Sophus (third party, CMakeLists.txt:46, no version pinned) is restated from its published so3.hpp:
// SO3::exp / expAndTheta, SO3::log / logAndTheta (with their small-angle series, epsilon 1e-10), inverse =
// conjugate, product = Hamilton product (Sophus' first-order renormalisation of a product — a factor
// 2/(1+|q|^2) that is 1 to rounding for unit inputs — is not replayed).  Quaternions are (x, y, z, w).
constexpr double kSophusEps = 1e-10;

template <typename T>
void quat_mul(const T a[4], const T b[4], T out[4]) {
    out[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    out[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    out[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
    out[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}

template <typename T>
void so3_exp(const T w[3], T q[4]) {
    const T theta_sq = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    T imag, real;
    if (val(theta_sq) < kSophusEps * kSophusEps) {
        const T theta_po4 = theta_sq * theta_sq;
        imag = T(0.5) - T(1.0 / 48.0) * theta_sq + T(1.0 / 3840.0) * theta_po4;
        real = T(1.0) - T(1.0 / 8.0) * theta_sq + T(1.0 / 384.0) * theta_po4;
    } else {
        const T theta = sqrt_(theta_sq), half = T(0.5) * theta;
        imag = sin_(half) / theta;
        real = cos_(half);
    }
    q[0] = imag * w[0];
    q[1] = imag * w[1];
    q[2] = imag * w[2];
    q[3] = real;
}

template <typename T>
void so3_log(const T q[4], T w[3]) {
    const T squared_n = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
    const T qw = q[3];
    T two_atan;
    if (val(squared_n) < kSophusEps * kSophusEps) {
        const T squared_w = qw * qw;
        two_atan = T(2.0) / qw - T(2.0 / 3.0) * squared_n / (qw * squared_w);
    } else {
        const T n = sqrt_(squared_n);
        if (std::abs(val(qw)) < kSophusEps)
            two_atan = (val(qw) > 0 ? T(M_PI) : T(-M_PI)) / n;
        else
            two_atan = T(2.0) * atan_(n / qw) / n;
    }
    for (int k = 0; k < 3; k++) w[k] = two_atan * q[k];
}

template <typename T>
T residual_functor_so3(const T *intr, const T (*rq)[4], const T (*tp)[3], const double *rb, const double *tb,
                       const double obs[2], const double lm[3], double radius, bool fisheye = false) {
    // :103-108  Qwb = r_cp0 * prod_j exp(beta_j log(r_cp{j-1}^-1 r_cp{j}));  rb = cumulative basis (3 values)
    T Q[4] = {rq[0][0], rq[0][1], rq[0][2], rq[0][3]};
    for (int j = 1; j <= 3; j++) {
        const T inv[4] = {-rq[j - 1][0], -rq[j - 1][1], -rq[j - 1][2], rq[j - 1][3]};
        T rel[4], d[3], e[4], nq[4];
        quat_mul(inv, rq[j], rel);
        so3_log(rel, d);
        for (int k = 0; k < 3; k++) d[k] = T(rb[j - 1]) * d[k];
        so3_exp(d, e);
        quat_mul(Q, e, nq);
        for (int k = 0; k < 4; k++) Q[k] = nq[k];
    }
    T t[3];
    for (int k = 0; k < 3; k++) t[k] = T(tb[0]) * tp[0][k] + T(tb[1]) * tp[1][k] + T(tb[2]) * tp[2][k] + T(tb[3]) * tp[3][k];
    // unDistort (:36-63) and the ray / plane intersection (:118-131), as in the quaternion functor
    T Xc[3];
    Xc[0] = (T(obs[0]) - intr[2]) / intr[0];
    Xc[1] = (T(obs[1]) - intr[3]) / intr[1];
    Xc[2] = T(1.0);
    const T r2 = Xc[0] * Xc[0] + Xc[1] * Xc[1];
    const T r4 = r2 * r2, r6 = r4 * r2, r8 = r6 * r2, r10 = r8 * r2;
    T coeff = T(1.0) + intr[4] * r2 + intr[5] * r4 + intr[6] * r6 + intr[7] * r8 + intr[8] * r10;
    if (fisheye) coeff = fisheye_scale(r2, coeff);
    Xc[0] *= coeff;
    Xc[1] *= coeff;
    const T tx = T(2.0) * Q[0], ty = T(2.0) * Q[1], tz = T(2.0) * Q[2];
    const T twx = tx * Q[3], twy = ty * Q[3], txx = tx * Q[0], txz = tz * Q[0], tyy = ty * Q[1], tyz = tz * Q[1];
    const T r2row[3] = {txz - twy, tyz + twx, T(1.0) - (txx + tyy)};
    const T depth = -t[2] / (r2row[0] * Xc[0] + r2row[1] * Xc[1] + r2row[2] * Xc[2]);
    for (int k = 0; k < 3; k++) Xc[k] *= depth;
    T uv[3] = {Q[1] * Xc[2] - Q[2] * Xc[1], Q[2] * Xc[0] - Q[0] * Xc[2], Q[0] * Xc[1] - Q[1] * Xc[0]};
    for (int k = 0; k < 3; k++) uv[k] = uv[k] + uv[k];
    const T c2[3] = {Q[1] * uv[2] - Q[2] * uv[1], Q[2] * uv[0] - Q[0] * uv[2], Q[0] * uv[1] - Q[1] * uv[0]};
    T Xw[3];
    for (int k = 0; k < 3; k++) Xw[k] = Xc[k] + Q[3] * uv[k] + c2[k] + t[k];
    const T d0 = Xw[0] - T(lm[0]), d1 = Xw[1] - T(lm[1]), d2 = Xw[2] - T(lm[2]);
    return sqrt_(d0 * d0 + d1 * d1 + d2 * d2) - T(radius);
}

}  // namespace

extern "C" {

// BsplineReal::findSpan (:208-231) for a clamped knot vector with n_cp control points, degree 3
uint32_t oracle_find_span(const double *knots, uint32_t n_cp, double u) {
    const size_t degree = 3;
    const size_t n = (n_cp + degree + 1) - 2 - degree;
    if (u == knots[n + 1]) return (uint32_t) n;
    size_t low = degree, high = n + 1, mid = (low + high) / 2;
    while (u < knots[mid] || u >= knots[mid + 1]) {
        if (u < knots[mid]) high = mid; else low = mid;
        mid = (low + high) / 2;
    }
    return (uint32_t) mid;
}

// BsplineReal::dersBasisFuns(u, span, 0, ders) -> ders[0][0..3]  (:107-145)
void oracle_basis(const double *knots, uint32_t span, double u, double *b4) {
    const int degree = 3;
    double ndu[4][4], left[4], right[4];
    ndu[0][0] = 1;
    for (int j = 1; j <= degree; j++) {
        left[j] = u - knots[span + 1 - j];
        right[j] = knots[span + j] - u;
        double saved = 0.0;
        for (int r = 0; r < j; ++r) {
            ndu[j][r] = right[r + 1] + left[j - r];
            double temp = ndu[r][j - 1] / ndu[j][r];
            ndu[r][j] = saved + right[r + 1] * temp;
            saved = left[j - r] * temp;
        }
        ndu[j][j] = saved;
    }
    for (int j = 0; j <= degree; j++) b4[j] = ndu[j][degree];
}

// residual and its Jacobian: J37 = ambient partials [intr 9 | q0..q3 (4 each) | t0..t3 (3 each)] as
// Ceres' autodiff would return them; J33 = after EigenQuaternionParameterization (tangent space)
// [intr 9 | delta_0..delta_3 (3 each) | t0..t3].  Either may be NULL.
double oracle_residual_cam(const double *intr, const double *q4x4, const double *t4x3, const double *basis4,
                           const double *obs2, const double *lm3, double radius, double *J37, double *J33, int fisheye);
double oracle_residual(const double *intr, const double *q4x4, const double *t4x3, const double *basis4,
                       const double *obs2, const double *lm3, double radius, double *J37, double *J33) {
    return oracle_residual_cam(intr, q4x4, t4x3, basis4, obs2, lm3, radius, J37, J33, 0);
}
double oracle_residual_cam(const double *intr, const double *q4x4, const double *t4x3, const double *basis4,
                           const double *obs2, const double *lm3, double radius, double *J37, double *J33, int fisheye) {
    Jet ji[9], jq[4][4], jt[4][3];
    for (int i = 0; i < 9; i++) ji[i] = Jet::var(intr[i], i);
    for (int j = 0; j < 4; j++) {
        for (int k = 0; k < 4; k++) jq[j][k] = Jet::var(q4x4[4 * j + k], 9 + 4 * j + k);
        for (int k = 0; k < 3; k++) jt[j][k] = Jet::var(t4x3[3 * j + k], 25 + 3 * j + k);
    }
    const Jet r = residual_functor<Jet>(ji, jq, jt, basis4, basis4, obs2, lm3, radius, fisheye != 0);
    if (J37) for (int i = 0; i < NP; i++) J37[i] = r.v[i];
    if (J33) {
        for (int i = 0; i < 9; i++) J33[i] = r.v[i];
        for (int j = 0; j < 4; j++) {
            const double *x = q4x4 + 4 * j;
            const double *g = r.v + 9 + 4 * j;
            // EigenQuaternionParameterization::ComputeJacobian (4x3, row-major, xyzw):
            //   [ w  z -y ; -z  w  x ;  y -x  w ; -x -y -z ]
            J33[9 + 3 * j + 0] = g[0] * x[3] - g[1] * x[2] + g[2] * x[1] - g[3] * x[0];
            J33[9 + 3 * j + 1] = g[0] * x[2] + g[1] * x[3] - g[2] * x[0] - g[3] * x[1];
            J33[9 + 3 * j + 2] = -g[0] * x[1] + g[1] * x[0] + g[2] * x[3] - g[3] * x[2];
            for (int k = 0; k < 3; k++) J33[21 + 3 * j + k] = r.v[25 + 3 * j + k];
        }
    }
    return r.a;
}

// SO3 variant: basis4 = the four N values; the cumulative basis of BsplineSO3::derBasisFuns
// (core/spline/src/BsplineSO3.cpp:88-94) is formed here.  J33 = ambient partials times
// Sophus::SO3::Dx_this_mul_exp_x_at_0 (LocalParameterizationSO3::ComputeJacobian, BsplineSO3.hpp:209-216):
// the tangent of  r_cp <- r_cp * exp(delta).
double oracle_residual_so3_cam(const double *intr, const double *q4x4, const double *t4x3, const double *basis4,
                               const double *obs2, const double *lm3, double radius, double *J37, double *J33, int fisheye);
double oracle_residual_so3(const double *intr, const double *q4x4, const double *t4x3, const double *basis4,
                           const double *obs2, const double *lm3, double radius, double *J37, double *J33) {
    return oracle_residual_so3_cam(intr, q4x4, t4x3, basis4, obs2, lm3, radius, J37, J33, 0);
}
double oracle_residual_so3_cam(const double *intr, const double *q4x4, const double *t4x3, const double *basis4,
                               const double *obs2, const double *lm3, double radius, double *J37, double *J33, int fisheye) {
    double beta[3];
    beta[2] = basis4[3];
    beta[1] = beta[2] + basis4[2];
    beta[0] = beta[1] + basis4[1];
    Jet ji[9], jq[4][4], jt[4][3];
    for (int i = 0; i < 9; i++) ji[i] = Jet::var(intr[i], i);
    for (int j = 0; j < 4; j++) {
        for (int k = 0; k < 4; k++) jq[j][k] = Jet::var(q4x4[4 * j + k], 9 + 4 * j + k);
        for (int k = 0; k < 3; k++) jt[j][k] = Jet::var(t4x3[3 * j + k], 25 + 3 * j + k);
    }
    const Jet r = residual_functor_so3<Jet>(ji, jq, jt, beta, basis4, obs2, lm3, radius, fisheye != 0);
    if (J37) for (int i = 0; i < NP; i++) J37[i] = r.v[i];
    if (J33) {
        for (int i = 0; i < 9; i++) J33[i] = r.v[i];
        for (int j = 0; j < 4; j++) {
            const double *x = q4x4 + 4 * j;
            const double *g = r.v + 9 + 4 * j;
            // Dx_this_mul_exp_x_at_0 (4x3, xyzw rows): 0.5 * [ w -z  y ;  z  w -x ; -y  x  w ; -x -y -z ]
            J33[9 + 3 * j + 0] = 0.5 * (g[0] * x[3] + g[1] * x[2] - g[2] * x[1] - g[3] * x[0]);
            J33[9 + 3 * j + 1] = 0.5 * (-g[0] * x[2] + g[1] * x[3] + g[2] * x[0] - g[3] * x[1]);
            J33[9 + 3 * j + 2] = 0.5 * (g[0] * x[1] - g[1] * x[0] + g[2] * x[3] - g[3] * x[2]);
            for (int k = 0; k < 3; k++) J33[21 + 3 * j + k] = r.v[25 + 3 * j + k];
        }
    }
    return r.a;
}

double oracle_residual_value(const double *intr, const double *q4x4, const double *t4x3, const double *basis4,
                             const double *obs2, const double *lm3, double radius) {
    double q[4][4], t[4][3];
    for (int j = 0; j < 4; j++) {
        for (int k = 0; k < 4; k++) q[j][k] = q4x4[4 * j + k];
        for (int k = 0; k < 3; k++) t[j][k] = t4x3[3 * j + k];
    }
    return residual_functor<double>(intr, q, t, basis4, basis4, obs2, lm3, radius);
}

// PinholeCamera::inverseRadialDistortion (PinholeCamera.cpp:69-95): (k1,k2,k3,k4) -> b1..b5
void oracle_inverse_radial(const double *k4, double *b5) {
    const double k1 = k4[0], k2 = k4[1], k3 = k4[2], k4_ = k4[3];
    b5[0] = -k1;
    b5[1] = 3 * k1 * k1 - k2;
    b5[2] = -12 * k1 * k1 * k1 + 8 * k1 * k2 - k3;
    b5[3] = 55 * k1 * k1 * k1 * k1 - 55 * k1 * k1 * k2 + 5 * k2 * k2 + 10 * k1 * k3 - k4_;
    b5[4] = -273 * k1 * k1 * k1 * k1 * k1 + 364 * k1 * k1 * k1 * k2 - 78 * k1 * k2 * k2 - 78 * k1 * k1 * k3 +
            12 * k2 * k3 + 12 * k1 * k4_;
}

// ---- whole-problem evaluation on the CPU (the solver cpu_baseline and the checker of the GPU
// normal equations).  Layout of a problem: intr[9]; n_cp control points (q [n_cp][4], t [n_cp][3])
// of ONE spline segment with knots[n_cp + 4]; M residual records: obs [M][2], time [M], lm id [M];
// landmarks [L][3].  Outputs (any may be NULL): cost = sum rho/2; g[9 + 6 n_cp] = J^T r in tangent
// order [intr | (delta_c, t_c) per control point]; H dense [(9+6n_cp)^2] row-major (small problems).
double oracle_evaluate_mode(const double *intr, uint32_t n_cp, const double *q, const double *t, const double *knots,
                            uint64_t M, const double *obs, const double *time, const uint32_t *lm_id,
                            const double *landmarks, double radius, double huber_a, int use_so3, double *g, double *H);

double oracle_evaluate(const double *intr, uint32_t n_cp, const double *q, const double *t, const double *knots,
                       uint64_t M, const double *obs, const double *time, const uint32_t *lm_id,
                       const double *landmarks, double radius, double huber_a, double *g, double *H) {
    return oracle_evaluate_mode(intr, n_cp, q, t, knots, M, obs, time, lm_id, landmarks, radius, huber_a, 0, g, H);
}

// use_so3 != 0: the cumulative SO3 spline (CalibReprojectionError_SO3 + LocalParameterizationSO3)
double oracle_evaluate_mode(const double *intr, uint32_t n_cp, const double *q, const double *t, const double *knots,
                            uint64_t M, const double *obs, const double *time, const uint32_t *lm_id,
                            const double *landmarks, double radius, double huber_a, int use_so3, double *g, double *H) {
    const size_t N = 9 + 6 * (size_t) n_cp;
    if (g) std::fill(g, g + N, 0.0);
    if (H) std::fill(H, H + N * N, 0.0);
    double cost = 0;
    for (uint64_t m = 0; m < M; m++) {
        const double u = time[m];
        const uint32_t span = oracle_find_span(knots, n_cp, u);
        double b[4], J[33];
        oracle_basis(knots, span, u, b);
        const uint32_t c0 = span - 3;
        // (use_so3: bit 0 = the SO3 spline, bit 1 = the fisheye camera)
        double r = ((use_so3 & 1) ? oracle_residual_so3_cam : oracle_residual_cam)(intr, q + 4 * (size_t) c0, t + 3 * (size_t) c0, b,
                                                                                   obs + 2 * m, landmarks + 3 * (size_t) lm_id[m], radius,
                                                                                   nullptr, (g || H) ? J : nullptr, (use_so3 >> 1) & 1);
        // HuberLoss + Corrector
        const double s = r * r, a2 = huber_a * huber_a;
        double rho, scale;
        if (s <= a2) {
            rho = s;
            scale = 1.0;
        } else {
            const double rt = std::sqrt(s);
            rho = 2 * huber_a * rt - a2;
            scale = std::sqrt(huber_a / rt);
        }
        cost += 0.5 * rho;
        if (!(g || H)) continue;
        r *= scale;
        size_t idx[33];
        for (int i = 0; i < 9; i++) idx[i] = i;
        for (int j = 0; j < 4; j++)
            for (int k = 0; k < 3; k++) {
                idx[9 + 3 * j + k] = 9 + 6 * (size_t) (c0 + j) + k;
                idx[21 + 3 * j + k] = 9 + 6 * (size_t) (c0 + j) + 3 + k;
            }
        for (int i = 0; i < 33; i++) J[i] *= scale;
        if (g) for (int i = 0; i < 33; i++) g[idx[i]] += J[i] * r;
        if (H) for (int i = 0; i < 33; i++) for (int j = 0; j < 33; j++) H[idx[i] * N + idx[j]] += J[i] * J[j];
    }
    return cost;
}

// The normal equations of a one-segment problem of ANY size, in the product's accumulation-buffer layout (the checker of the
// GPU normal equations at benchmark size: oracle_evaluate_mode's dense H is (9 + 6 n_cp)^2 doubles).  acc[91 + 204 n_cp]:
// [0] cost = sum rho / 2 | [1..9] g_intr | [10 + 9 i + j], j >= i: H_intr upper | per control point c at 91 + 204 c:
// g_c[6] (rot 3, trans 3) | H_c,intr[6][9] | H_c,c+d[6][6] for d = 0..3 (d = 0: upper triangle only) — the sums Ceres'
// normal-equation build forms from the autodiff rows (EventCalibSpline.cpp:196-247), here from the dual-number rows above.
// n_threads workers over contiguous ranges of the residuals, each into its own buffer, added in thread order.
void oracle_evaluate_arrow_mt(const double *intr, uint32_t n_cp, const double *q, const double *t, const double *knots,
                              uint64_t M, const double *obs, const double *time, const uint32_t *lm_id, const double *landmarks,
                              double radius, double huber_a, int mode, int n_threads, double *acc) {
    const size_t HEAD = 91, PER = 204, N = HEAD + PER * (size_t) n_cp;
    if (n_threads < 1) n_threads = 1;
    std::vector<std::vector<double>> part((size_t) n_threads);
    auto work = [&](int w) {
        std::vector<double> &a = part[(size_t) w];
        a.assign(N, 0.0);
        const uint64_t lo = M * (uint64_t) w / (uint64_t) n_threads, hi = M * (uint64_t) (w + 1) / (uint64_t) n_threads;
        for (uint64_t m = lo; m < hi; m++) {
            const double u = time[m];
            const uint32_t span = oracle_find_span(knots, n_cp, u);
            double b[4], J[33];
            oracle_basis(knots, span, u, b);
            const uint32_t c0 = span - 3;
            double r = ((mode & 1) ? oracle_residual_so3_cam : oracle_residual_cam)(intr, q + 4 * (size_t) c0, t + 3 * (size_t) c0, b, obs + 2 * m,
                                                                                    landmarks + 3 * (size_t) lm_id[m], radius, nullptr, J,
                                                                                    (mode >> 1) & 1);
            const double s = r * r, a2 = huber_a * huber_a;   // HuberLoss + Corrector, as oracle_evaluate_mode
            double rho, scale;
            if (s <= a2) {
                rho = s;
                scale = 1.0;
            } else {
                const double rt = std::sqrt(s);
                rho = 2 * huber_a * rt - a2;
                scale = std::sqrt(huber_a / rt);
            }
            a[0] += 0.5 * rho;
            r *= scale;
            for (int i = 0; i < 33; i++) J[i] *= scale;
            for (int i = 0; i < 9; i++) {
                a[1 + i] += J[i] * r;
                for (int j = i; j < 9; j++) a[10 + 9 * i + j] += J[i] * J[j];
            }
            double v[4][6];   // the row's entries of control point c0 + j: rotation tangent, translation
            for (int j = 0; j < 4; j++)
                for (int k = 0; k < 3; k++) {
                    v[j][k] = J[9 + 3 * j + k];
                    v[j][3 + k] = J[21 + 3 * j + k];
                }
            for (int j = 0; j < 4; j++) {
                double *rec = a.data() + HEAD + PER * (size_t) (c0 + j);
                for (int k = 0; k < 6; k++) {
                    rec[k] += v[j][k] * r;
                    for (int i = 0; i < 9; i++) rec[6 + 9 * k + i] += v[j][k] * J[i];
                }
                for (int d = 0; j + d < 4; d++)
                    for (int ka = 0; ka < 6; ka++)
                        for (int kb = (d == 0 ? ka : 0); kb < 6; kb++) rec[60 + 36 * d + 6 * ka + kb] += v[j][ka] * v[j + d][kb];
            }
        }
    };
    std::vector<std::thread> th;
    for (int w = 1; w < n_threads; w++) th.emplace_back(work, w);
    work(0);
    for (auto &x : th) x.join();
    std::fill(acc, acc + N, 0.0);
    for (int w = 0; w < n_threads; w++)
        for (size_t i = 0; i < N; i++) acc[i] += part[(size_t) w][i];
}

}  // extern "C"
