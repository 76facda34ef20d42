// Per-event reprojection residual of the continuous-time calibration and its analytic gradient.
//
// Replaces the Ceres autodiff functor CalibReprojectionError::operator()
// (event_camera_calib/include/opengv2/event_camera_calib/EventCalibSpline.hpp:158-229, unDistort
// :36-63) together with ceres::EigenQuaternionParameterization (used at EventCalibSpline.cpp:116-135)
// and the degree-3 B-spline basis of core/spline/include/opengv2/spline/BsplineReal.hpp:107-145,
// 208-231 (NURBS book A2.1/A2.2).  Plain functions, usable from HIP kernels and host C++.
//
// Residual (B.2 of SURVEY): v = sum_j b_j q_j (xyzw), q = v/|v|, T = sum_j b_j t_j;
//   x = (u-cx)/fx, y = (v-cy)/fy, r2 = x^2+y^2, c = 1 + k1 r2 + .. + k5 r2^5, p = (x c, y c, 1);
//   Y = R(q) p, Xw = T - T_z * Y / Y_z   (ray / plane z = 0 intersection: depth = -T_z / (R3 . p));
//   res = |Xw - lm| - Rc.
// FISHEYE (BASELINE configs[4]; new functionality — the reference's solver takes the radial model only,
// EventCalibSpline.cpp:97-99): the Kannala-Brandt camera in the SAME inverse form, so that the nine intrinsics keep their
// slots and PinholeCamera::inverseRadialDistortion (a plain series reversion) initialises them from the forward
// coefficients of cv::fisheye::calibrate: with r = sqrt(x^2 + y^2) the DISTORTED angle theta_d of the pixel,
//   theta = r (1 + k1 r^2 + .. + k5 r^10)   (the inverse polynomial),   p = (x, y, 0) tan(theta) / r + (0, 0, 1),
// i.e. c = tan(theta) / r takes the place of the radial factor; everything behind p is shared.
// The Jacobian row has 33 entries in tangent space:
//   [0..8]   intrinsics fx fy cx cy k1..k5
//   [9+3j..] rotation control point j (j = 0..3): delta of  q_j <- exp(delta) (x) q_j
//   [21+3j..] translation control point j
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define ECAL_HD __host__ __device__ __forceinline__
#else
#define ECAL_HD inline
#endif

namespace ecal {

constexpr int RES_NJ = 33;  // tangent-space width of one residual's Jacobian row

// 1 / sqrt(x): on the device the reciprocal square root (v_rsq_f64 + refinement, ~1 ulp) instead of a square root AND a division
// (~27 FP64 instructions of the ~440 a batch of residuals costs; the kernel is bound by FP64 issue, design/08_solver.md)
ECAL_HD double res_rsqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return rsqrt(x);
#else
    return 1.0 / sqrt(x);
#endif
}

// knot span index for u in a clamped knot vector with n_cp control points, degree 3
// (BsplineReal.hpp:208-231; u == last knot -> last span)
ECAL_HD uint32_t spline_find_span(const double *knots, uint32_t n_cp, double u) {
    const uint32_t n = n_cp - 1;
    if (u == knots[n + 1]) return n;
    uint32_t low = 3, high = n + 1, mid = (low + high) / 2;
    while (u < knots[mid] || u >= knots[mid + 1]) {
        if (u < knots[mid]) high = mid; else low = mid;
        mid = (low + high) / 2;
    }
    return mid;
}

// the four non-zero cubic basis functions N_{span-3..span,3}(u) (BsplineReal.hpp:107-145, ders[0])
ECAL_HD void spline_basis(const double *knots, uint32_t span, double u, double b[4]) {
    double left[4], right[4], ndu[4][4];
    ndu[0][0] = 1.0;
    for (int j = 1; j <= 3; j++) {
        left[j] = u - knots[span + 1 - j];
        right[j] = knots[span + j] - u;
        double saved = 0.0;
        for (int r = 0; r < j; r++) {
            ndu[j][r] = right[r + 1] + left[j - r];
            const double temp = ndu[r][j - 1] / ndu[j][r];
            ndu[r][j] = saved + right[r + 1] * temp;
            saved = left[j - r] * temp;
        }
        ndu[j][j] = saved;
    }
    for (int j = 0; j <= 3; j++) b[j] = ndu[j][3];
}

// The six denominators of the cubic basis recursion are knot differences — constants of the span, not of u:
// ndu[j][r] = right[r+1] + left[j-r] = knots[span+r+1] - knots[span+1-j+r].  A chunk of residuals shares its span, so
// the kernel takes their reciprocals once and the per-residual basis costs no division.
ECAL_HD void spline_span_inverses(const double *knots, uint32_t span, double inv[6]) {
    int o = 0;
    for (int j = 1; j <= 3; j++)
        for (int r = 0; r < j; r++) inv[o++] = 1.0 / (knots[span + r + 1] - knots[span + 1 - j + r]);
}
ECAL_HD void spline_basis_inv(const double *knots, uint32_t span, const double inv[6], double u, double b[4]) {
    double left[4], right[4], N[4];
    N[0] = 1.0;
    int o = 0;
    for (int j = 1; j <= 3; j++) {
        left[j] = u - knots[span + 1 - j];
        right[j] = knots[span + j] - u;
        double saved = 0.0;
        for (int r = 0; r < j; r++) {
            const double temp = N[r] * inv[o++];
            N[r] = saved + right[r + 1] * temp;
            saved = left[j - r] * temp;
        }
        N[j] = saved;
    }
    for (int j = 0; j <= 3; j++) b[j] = N[j];
}

ECAL_HD double huber_scale(double r, double a, double *half_rho, double inv_a);

struct ResidualInput {
    // sc_out != nullptr: the core applies the robust loss (Huber, huber_a) itself — *sc_out = sqrt(rho') and *half_rho_out = rho / 2 of the residual it
    // returns, and the Jacobian row comes out ALREADY scaled by sqrt(rho') (the row is linear in d res / d Xw, which is scaled at
    // its root: three multiplications instead of 33 on the finished row)
    double huber_a = 0.0, inv_huber_a = 0.0;
    double *sc_out = nullptr, *half_rho_out = nullptr;
    bool dead = false;    // (with sc_out) a row that must come out as zeros: *sc_out = 0 (a lane past the end of its chunk)
    double u, v;          // observed pixel
    double lmx, lmy, lmz; // landmark (circle centre on the board, z = 0)
    double radius;        // circle radius (world units)
    double b[4];          // basis values (same for the rotation and the translation spline)
    double ifx, ify;      // 1 / fx, 1 / fy (the kernel hoists them out of its residual loop); 0 = compute here
};

// Shared tail of both rotation parameterisations: residual for the unit quaternion (ux,uy,uz,w) and translation T;
// when J != nullptr fills J[0..8] (intrinsics), gq = d res / d (unit quaternion, ambient xyzw) and gT = d res / d T.
template <bool FISHEYE = false>
ECAL_HD double residual_core(const ResidualInput &in, const double *intr, double ux, double uy, double uz, double w,
                             const double T[3], double *J, double gq[4], double gT[3]) {
    const double cx = intr[2], cy = intr[3];
    // divisions are reciprocal + multiply: one rcp each for fx, fy (hoisted by the kernel), Y_z, |Xw - lm|
    const double ifx = in.ifx != 0.0 ? in.ifx : 1.0 / intr[0], ify = in.ify != 0.0 ? in.ify : 1.0 / intr[1];
    // undistorted ray
    const double x = (in.u - cx) * ifx, y = (in.v - cy) * ify;
    const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2, r8 = r6 * r2, r10 = r8 * r2;
    const double poly = 1.0 + intr[4] * r2 + intr[5] * r4 + intr[6] * r6 + intr[7] * r8 + intr[8] * r10;
    double c = poly, sec2 = 1.0;   // sec2 = d c / d poly
    if (FISHEYE) {
        if (r2 > 1e-16) {
            const double r = sqrt(r2), tn = tan(r * poly);
            sec2 = 1.0 + tn * tn;
            c = tn / r;
        }   // (r -> 0: tan(r poly) / r -> poly, sec^2 -> 1)
    }
    const double px = x * c, py = y * c, pz = 1.0;
    // Y = p + 2 w (u x p) + 2 (u (u.p) - p (u.u))
    const double cxp0 = uy * pz - uz * py, cxp1 = uz * px - ux * pz, cxp2 = ux * py - uy * px;  // u x p
    const double udp = ux * px + uy * py + uz * pz, udu = ux * ux + uy * uy + uz * uz;
    const double Y0 = px + 2 * w * cxp0 + 2 * (ux * udp - px * udu);
    const double Y1 = py + 2 * w * cxp1 + 2 * (uy * udp - py * udu);
    const double Y2 = pz + 2 * w * cxp2 + 2 * (uz * udp - pz * udu);
    const double iY2 = 1.0 / Y2;
    const double s = -T[2] * iY2;  // depth
    const double Xw0 = T[0] + s * Y0, Xw1 = T[1] + s * Y1, Xw2 = T[2] + s * Y2;
    const double d0 = Xw0 - in.lmx, d1 = Xw1 - in.lmy, d2 = Xw2 - in.lmz;
    const double dd = d0 * d0 + d1 * d1 + d2 * d2;
    const double idist = res_rsqrt(dd);
    const double dist = dd * idist;
    const double res = dist - in.radius;
    if (!J) return res;

    double esc = idist;
    if (in.sc_out) {
        const double sc = in.dead ? 0.0 : huber_scale(res, in.huber_a, in.half_rho_out, in.inv_huber_a);
        *in.sc_out = sc;
        esc = idist * sc;
    }
    const double e0 = d0 * esc, e1 = d1 * esc, e2 = d2 * esc;  // d res / d Xw (times sqrt(rho') under the robust loss)
    const double eY = e0 * Y0 + e1 * Y1 + e2 * Y2;
    // Xw = T - T_z Y / Y_z
    gT[0] = e0;
    gT[1] = e1;
    gT[2] = e2 - eY * iY2;
    const double k = s;
    const double gY0 = k * e0, gY1 = k * e1, gY2 = k * e2 - k * eY * iY2;
    // d Y / d p = R(q) = I + 2 w [u]x + 2 (u u^T - (u.u) I);   g_p = R^T g_Y = g_Y - 2 w (u x g_Y) + 2 (u (u.g_Y) - g_Y (u.u))
    const double udg = ux * gY0 + uy * gY1 + uz * gY2;
    const double cxg0 = uy * gY2 - uz * gY1, cxg1 = uz * gY0 - ux * gY2;
    const double gp0 = gY0 - 2 * w * cxg0 + 2 * (ux * udg - gY0 * udu);
    const double gp1 = gY1 - 2 * w * cxg1 + 2 * (uy * udg - gY1 * udu);
    // intrinsics
    double cp = intr[4] + 2 * intr[5] * r2 + 3 * intr[6] * r4 + 4 * intr[7] * r6 + 5 * intr[8] * r8;  // d poly / d r2
    if (FISHEYE) {
        // c = tan(r poly) / r:  dc/dr2 = (sec^2 (poly / 2 + r2 poly') - c / 2) / r2  ->  poly' + poly^3 / 3 at r = 0
        cp = r2 > 1e-8 ? (sec2 * (0.5 * poly + r2 * cp) - 0.5 * c) / r2 : cp + poly * poly * poly * (1.0 / 3.0);
    }
    const double gx = gp0 * (c + 2 * x * x * cp) + gp1 * (2 * x * y * cp);
    const double gy = gp0 * (2 * x * y * cp) + gp1 * (c + 2 * y * y * cp);
    J[0] = -gx * x * ifx;
    J[1] = -gy * y * ify;
    J[2] = -gx * ifx;
    J[3] = -gy * ify;
    const double gk = (gp0 * x + gp1 * y) * sec2;   // d c / d k_i = sec^2 r2^i (pinhole: sec2 = 1)
    J[4] = gk * r2;
    J[5] = gk * r4;
    J[6] = gk * r6;
    J[7] = gk * r8;
    J[8] = gk * r10;
    // unit quaternion: g_w = 2 (u x p).g_Y ;  g_u = 2 w (p x g_Y) + 2 ((u.p) g_Y + p (u.g_Y) - 2 u (p.g_Y))
    const double pdg = px * gY0 + py * gY1 + pz * gY2;
    const double pxg0 = py * gY2 - pz * gY1, pxg1 = pz * gY0 - px * gY2, pxg2 = px * gY1 - py * gY0;
    gq[0] = 2 * w * pxg0 + 2 * (udp * gY0 + px * udg - 2 * ux * pdg);
    gq[1] = 2 * w * pxg1 + 2 * (udp * gY1 + py * udg - 2 * uy * pdg);
    gq[2] = 2 * w * pxg2 + 2 * (udp * gY2 + pz * udg - 2 * uz * pdg);
    gq[3] = 2 * (cxp0 * gY0 + cxp1 * gY1 + cxp2 * gY2);
    return res;
}

// intr[9]; q[4][4] rotation control points (x y z w); t[4][3] translation control points.
// Returns the residual; if J != nullptr fills the 33 tangent-space partials.
template <bool FISHEYE = false>
ECAL_HD double spline_residual(const ResidualInput &in, const double *intr, const double (*q)[4], const double (*t)[3],
                               double *J) {
    // pose at the event time
    double vq[4] = {0, 0, 0, 0}, T[3] = {0, 0, 0};
    for (int j = 0; j < 4; j++) {
        for (int k = 0; k < 4; k++) vq[k] += in.b[j] * q[j][k];
        for (int k = 0; k < 3; k++) T[k] += in.b[j] * t[j][k];
    }
    const double ivn = res_rsqrt(vq[0] * vq[0] + vq[1] * vq[1] + vq[2] * vq[2] + vq[3] * vq[3]);
    const double ux = vq[0] * ivn, uy = vq[1] * ivn, uz = vq[2] * ivn, w = vq[3] * ivn;
    double gq[4], gT[3];
    const double res = residual_core<FISHEYE>(in, intr, ux, uy, uz, w, T, J, gq, gT);
    if (!J) return res;
    // through the normalisation q = v / |v|
    const double qdg = ux * gq[0] + uy * gq[1] + uz * gq[2] + w * gq[3];
    const double gv[4] = {(gq[0] - ux * qdg) * ivn, (gq[1] - uy * qdg) * ivn, (gq[2] - uz * qdg) * ivn,
                          (gq[3] - w * qdg) * ivn};
    for (int j = 0; j < 4; j++) {
        const double bj = in.b[j];
        // EigenQuaternionParameterization: q_j <- exp(delta) (x) q_j ; d/d delta at 0 (4x3, xyzw rows)
        //   [ w  z -y ; -z  w  x ;  y -x  w ; -x -y -z ]
        const double qx = q[j][0], qy = q[j][1], qz = q[j][2], qw = q[j][3];
        J[9 + 3 * j + 0] = bj * (gv[0] * qw - gv[1] * qz + gv[2] * qy - gv[3] * qx);
        J[9 + 3 * j + 1] = bj * (gv[0] * qz + gv[1] * qw - gv[2] * qx - gv[3] * qy);
        J[9 + 3 * j + 2] = bj * (-gv[0] * qy + gv[1] * qx + gv[2] * qw - gv[3] * qz);
        J[21 + 3 * j + 0] = bj * gT[0];
        J[21 + 3 * j + 1] = bj * gT[1];
        J[21 + 3 * j + 2] = bj * gT[2];
    }
    return res;
}

// ---- cumulative SO3 spline (useSO3 = 1) ------------------------------------------------------------------------
// CalibReprojectionError_SO3::operator() (EventCalibSpline.hpp:65-135) with LocalParameterizationSO3
// (core/spline/include/opengv2/spline/BsplineSO3.hpp:190-221: r_cp <- r_cp * exp(delta)):
//   R = R_0 A_1 A_2 A_3,  A_j = Exp(beta_j d_j),  d_j = Log(R_{j-1}^T R_j),  beta = cumulative basis
//   (BsplineSO3.cpp:88-94: beta_3 = N_3, beta_2 = N_3 + N_2, beta_1 = N_3 + N_2 + N_1).
// Analytic tangent Jacobian instead of the reference's autodiff through Sophus: with g_w = d res / d omega for
// R <- R Exp(omega),  v_3 = g_w, v_{j-1} = A_j v_j (so v_0 = A_1 A_2 A_3 g_w),  m_j = beta_j Jl(beta_j d_j) v_j,
//   d res / d delta_0 = v_0 - Jr^-1(d_1) m_1,   d res / d delta_i = Jl^-1(d_i) m_i - Jr^-1(d_{i+1}) m_{i+1},
//   d res / d delta_3 = Jl^-1(d_3) m_3
// (Jr/Jl = right/left Jacobians of SO(3), Jl(x) = Jr(x)^T).  Rotations are handled as rotation vectors / matrices
// applied to vectors; quaternions are (x, y, z, w), q and -q denote the same rotation.

ECAL_HD void so3_cross(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

// out = (I + a [phi]x + b [phi]x^2) v
ECAL_HD void so3_apply(double a, double b, const double phi[3], const double v[3], double out[3]) {
    double c1[3], c2[3];
    so3_cross(phi, v, c1);
    so3_cross(phi, c1, c2);
    for (int k = 0; k < 3; k++) out[k] = v[k] + a * c1[k] + b * c2[k];
}

// Exp(phi) v  (Rodrigues)
ECAL_HD void so3_rotate(const double phi[3], const double v[3], double out[3]) {
    const double t2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
    double a, b;
    if (t2 < 1e-12) {
        a = 1.0 - t2 / 6.0;
        b = 0.5 - t2 / 24.0;
    } else {
        const double th = sqrt(t2);
        a = sin(th) / th;
        b = (1.0 - cos(th)) / t2;
    }
    so3_apply(a, b, phi, v, out);
}

// Jl(phi) v = (I + (1 - cos t)/t^2 [phi]x + (t - sin t)/t^3 [phi]x^2) v
ECAL_HD void so3_jl(const double phi[3], const double v[3], double out[3]) {
    const double t2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
    double a, b;
    if (t2 < 1e-8) {
        a = 0.5 - t2 / 24.0;
        b = 1.0 / 6.0 - t2 / 120.0;
    } else {
        const double th = sqrt(t2);
        a = (1.0 - cos(th)) / t2;
        b = (th - sin(th)) / (t2 * th);
    }
    so3_apply(a, b, phi, v, out);
}

// Jl^-1(phi) v (sign = +1) or Jr^-1(phi) v (sign = -1):  (I -/+ 1/2 [phi]x + (1/t^2 - (1 + cos t)/(2 t sin t)) [phi]x^2) v
ECAL_HD void so3_jinv(const double phi[3], double sign, const double v[3], double out[3]) {
    const double t2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
    double b;
    if (t2 < 1e-8) {
        b = 1.0 / 12.0 + t2 / 720.0;
    } else {
        const double th = sqrt(t2);
        b = 1.0 / t2 - (1.0 + cos(th)) / (2.0 * th * sin(th));
    }
    so3_apply(-0.5 * sign, b, phi, v, out);
}

// Hamilton product (xyzw)
ECAL_HD void quat_mul(const double a[4], const double b[4], double o[4]) {
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}

// Sophus SO3::log of a unit quaternion: rotation vector with angle 2 atan(|v| / w) (in (-pi, pi))
ECAL_HD void so3_log(const double q[4], double phi[3]) {
    const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2], w = q[3];
    double f;
    if (n2 < 1e-20) {
        f = 2.0 / w - (2.0 / 3.0) * n2 / (w * w * w);
    } else {
        const double n = sqrt(n2);
        f = fabs(w) < 1e-10 ? (w > 0 ? M_PI : -M_PI) / n : 2.0 * atan(n / w) / n;
    }
    phi[0] = f * q[0];
    phi[1] = f * q[1];
    phi[2] = f * q[2];
}

// Sophus SO3::exp: rotation vector -> unit quaternion
ECAL_HD void so3_exp(const double phi[3], double q[4]) {
    const double t2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
    double im, re;
    if (t2 < 1e-20) {
        im = 0.5 - t2 / 48.0;
        re = 1.0 - t2 / 8.0;
    } else {
        const double th = sqrt(t2);
        im = sin(0.5 * th) / th;
        re = cos(0.5 * th);
    }
    q[0] = im * phi[0];
    q[1] = im * phi[1];
    q[2] = im * phi[2];
    q[3] = re;
}

// q[4][4]: SO3 control points as unit quaternions (x y z w); in.b = the four N values (the cumulative basis is
// formed here).  J layout as spline_residual, rotation columns = delta of r_cp_j <- r_cp_j * exp(delta).
template <bool FISHEYE = false>
ECAL_HD double spline_residual_so3(const ResidualInput &in, const double *intr, const double (*q)[4],
                                   const double (*t)[3], double *J) {
    double beta[3];
    beta[2] = in.b[3];
    beta[1] = beta[2] + in.b[2];
    beta[0] = beta[1] + in.b[1];
    double d[3][3], bd[3][3], Q[4] = {q[0][0], q[0][1], q[0][2], q[0][3]}, T[3] = {0, 0, 0};
    for (int j = 1; j <= 3; j++) {
        const double inv[4] = {-q[j - 1][0], -q[j - 1][1], -q[j - 1][2], q[j - 1][3]};
        double rel[4], e[4], nq[4];
        quat_mul(inv, q[j], rel);
        so3_log(rel, d[j - 1]);
        for (int k = 0; k < 3; k++) bd[j - 1][k] = beta[j - 1] * d[j - 1][k];
        so3_exp(bd[j - 1], e);
        quat_mul(Q, e, nq);
        for (int k = 0; k < 4; k++) Q[k] = nq[k];
    }
    for (int j = 0; j < 4; j++)
        for (int k = 0; k < 3; k++) T[k] += in.b[j] * t[j][k];
    double gq[4], gT[3];
    const double res = residual_core<FISHEYE>(in, intr, Q[0], Q[1], Q[2], Q[3], T, J, gq, gT);
    if (!J) return res;
    // g_w: R <- R Exp(omega) is Q <- Q (x) (omega/2, 1)
    double v[3], m[3], a[3], n[3];
    v[0] = 0.5 * (gq[0] * Q[3] + gq[1] * Q[2] - gq[2] * Q[1] - gq[3] * Q[0]);
    v[1] = 0.5 * (-gq[0] * Q[2] + gq[1] * Q[3] + gq[2] * Q[0] - gq[3] * Q[1]);
    v[2] = 0.5 * (gq[0] * Q[1] - gq[1] * Q[0] + gq[2] * Q[3] - gq[3] * Q[2]);
    double p_next[3] = {0, 0, 0};  // Jl^-1(d_{j}) m_{j} of the factor processed last (j = 3 first)
    for (int j = 3; j >= 1; j--) {
        so3_jl(bd[j - 1], v, m);
        for (int k = 0; k < 3; k++) m[k] *= beta[j - 1];
        so3_jinv(d[j - 1], +1.0, m, a);   // Jl^-1(d_j) m_j  -> column block of control point j
        so3_jinv(d[j - 1], -1.0, m, n);   // Jr^-1(d_j) m_j  -> subtracted from control point j-1
        for (int k = 0; k < 3; k++) J[9 + 3 * j + k] = a[k] - (j < 3 ? p_next[k] : 0.0);
        for (int k = 0; k < 3; k++) p_next[k] = n[k];
        so3_rotate(bd[j - 1], v, a);      // v_{j-1} = A_j v_j
        for (int k = 0; k < 3; k++) v[k] = a[k];
    }
    for (int k = 0; k < 3; k++) J[9 + k] = v[k] - p_next[k];
    for (int j = 0; j < 4; j++)
        for (int k = 0; k < 3; k++) J[21 + 3 * j + k] = in.b[j] * gT[k];
    return res;
}

// ceres::HuberLoss(a) + Corrector for a scalar residual (Ceres 1.x loss_function.cc / corrector.cc):
// s = r^2; rho(s) = s (s <= a^2) or 2 a sqrt(s) - a^2; rho'' <= 0, so residual and Jacobian row are
// both scaled by sqrt(rho'); cost contribution = rho / 2.
// Branch-free: |r| is sqrt(s), and sqrt(a / |r|) one reciprocal square root of |r| / a (inv_a = 1 / a: a constant of the call).
ECAL_HD double huber_scale(double r, double a, double *half_rho, double inv_a) {
    const double s = r * r, b = a * a, rt = fabs(r);
    const bool tail = !(s <= b);
    *half_rho = tail ? a * rt - 0.5 * b : 0.5 * s;
    const double q = rt * (inv_a != 0.0 ? inv_a : 1.0 / a);
    return tail ? res_rsqrt(q > 0 ? q : 1.0) : 1.0;   // (tail implies |r| > a >= 0: q > 0; Ceres clamps a / |r| with DBL_MIN)
}

// q <- q (x) exp(delta)   (LocalParameterizationSO3::Plus, BsplineSO3.hpp:196-203)
ECAL_HD void so3_plus(const double q[4], const double d[3], double out[4]) {
    double e[4];
    so3_exp(d, e);
    quat_mul(q, e, out);
}

// q <- exp(delta) (x) q   (EigenQuaternionParameterization::Plus, xyzw storage)
ECAL_HD void quaternion_plus(const double q[4], const double d[3], double out[4]) {
    const double nd = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    double dq[4];
    if (nd > 0.0) {
        const double sn = sin(nd) / nd;
        dq[0] = sn * d[0];
        dq[1] = sn * d[1];
        dq[2] = sn * d[2];
        dq[3] = cos(nd);
    } else {
        dq[0] = dq[1] = dq[2] = 0.0;
        dq[3] = 1.0;
    }
    // Hamilton product dq (x) q
    out[0] = dq[3] * q[0] + dq[0] * q[3] + dq[1] * q[2] - dq[2] * q[1];
    out[1] = dq[3] * q[1] - dq[0] * q[2] + dq[1] * q[3] + dq[2] * q[0];
    out[2] = dq[3] * q[2] + dq[0] * q[1] - dq[1] * q[0] + dq[2] * q[3];
    out[3] = dq[3] * q[3] - dq[0] * q[0] - dq[1] * q[1] - dq[2] * q[2];
}

}  // namespace ecal
