// Camera models of the init calibration: projection of a board point and its analytic Jacobian.
//
// The reference delegates this arithmetic to OpenCV (cv::projectPoints / cv::fisheye::projectPoints with their
// Jacobian outputs inside cv::calibrateCamera / cv::fisheye::calibrate, call sites
// event_camera_calib/src/EventCalibIni.cpp:127-130,188,198-199); OpenCV is third party and not in the tree, so the
// models are restated from its published formulas and checked against central differences of
// oracle/calib_oracle.py::project (tests/test_gpu_calib.py).
//
// Intrinsics slots (CB_NI = 12 doubles):
//   model 0 (pinhole, rational radial + tangential): fx fy cx cy k1 k2 p1 p2 k3 k4 k5 k6
//   model 1 (fisheye, Kannala-Brandt):                fx fy cx cy alpha k1 k2 k3 k4 - - -
// Pose: rvec (Rodrigues vector, cv::Rodrigues) and tvec.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ecal {

constexpr int CB_NI = 12;
constexpr int CB_NV = 6;
constexpr int CB_NP = CB_NI + CB_NV;  // Jacobian columns of one residual row
constexpr int CB_NC = CB_NP + 1;      // + the residual itself

struct RodriguesTerms {
    double R[9];
    double r[3];   // unit axis (0 when theta < eps)
    double a, b;   // sin(theta)/theta, (1 - cos(theta))/theta
};

__device__ __forceinline__ void rodrigues_terms(const double *v, RodriguesTerms &T) {
    const double th2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    const double th = sqrt(th2);
    if (th < 2.220446049250313e-16) {
        T.R[0] = T.R[4] = T.R[8] = 1;
        T.R[1] = T.R[2] = T.R[3] = T.R[5] = T.R[6] = T.R[7] = 0;
        T.r[0] = T.r[1] = T.r[2] = 0;
        T.a = 1;
        T.b = 0;
        return;
    }
    const double it = 1.0 / th;
    const double rx = v[0] * it, ry = v[1] * it, rz = v[2] * it;
    const double s = sin(th), c = cos(th);
    const double sh = sin(0.5 * th);
    const double c1 = 2 * sh * sh;  // 1 - cos(theta) without cancellation
    T.R[0] = c + c1 * rx * rx;
    T.R[1] = c1 * rx * ry - s * rz;
    T.R[2] = c1 * rx * rz + s * ry;
    T.R[3] = c1 * rx * ry + s * rz;
    T.R[4] = c + c1 * ry * ry;
    T.R[5] = c1 * ry * rz - s * rx;
    T.R[6] = c1 * rx * rz - s * ry;
    T.R[7] = c1 * ry * rz + s * rx;
    T.R[8] = c + c1 * rz * rz;
    T.r[0] = rx;
    T.r[1] = ry;
    T.r[2] = rz;
    T.a = s * it;
    T.b = c1 * it;
}

__device__ __forceinline__ void cross3(const double *a, const double *b, double *o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

// d(R M)/d rvec_i = r_i (r x Y) + [a (e_i - r_i r) + b (r x e_i)] x Y   with Y = R M
// (the Rodrigues derivative in the compact form of Gallego & Yezzi 2015, rewritten so that nothing divides by theta^2)
__device__ __forceinline__ void drot_point(const RodriguesTerms &T, const double *Y, double (&dY)[3][3]) {
    double rxY[3];
    cross3(T.r, Y, rxY);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        double e[3] = {0, 0, 0};
        e[i] = 1;
        double rxe[3], w[3], wxY[3];
        cross3(T.r, e, rxe);
#pragma unroll
        for (int k = 0; k < 3; k++) w[k] = T.a * (e[k] - T.r[i] * T.r[k]) + T.b * rxe[k];
        cross3(w, Y, wxY);
#pragma unroll
        for (int k = 0; k < 3; k++) dY[i][k] = T.r[i] * rxY[k] + wxY[k];
    }
}

// Projection of camera-frame point Y.  Outputs pixel (u, v); when JAC: du[18], dv[18] = derivatives w.r.t. the 12
// intrinsics slots (raw, before masking / aspect tie) — the pose columns are filled by the caller from P (2x3 =
// d(u,v)/dY).
template <bool JAC>
__device__ __forceinline__ void project_cam(int model, const double *in, const double *Y, double *u, double *v, double *du,
                                            double *dv, double (&P)[2][3]) {
    const double iz = 1.0 / Y[2];
    const double x = Y[0] * iz, y = Y[1] * iz;
    double xd, yd, Gxx, Gxy, Gyx, Gyy;  // G = d(xd, yd)/d(x, y)
    const double fx = in[0], fy = in[1];
    double alpha = 0;
    if (model == 0) {
        const double k1 = in[4], k2 = in[5], p1 = in[6], p2 = in[7], k3 = in[8], k4 = in[9], k5 = in[10], k6 = in[11];
        const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
        const double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
        const double cd = 1 + k1 * r2 + k2 * r4 + k3 * r6;
        const double icd = 1.0 / (1 + k4 * r2 + k5 * r4 + k6 * r6);
        const double g = cd * icd;
        xd = x * g + p1 * a1 + p2 * a2;
        yd = y * g + p1 * a3 + p2 * a1;
        if (JAC) {
            const double gp = (k1 + 2 * k2 * r2 + 3 * k3 * r4) * icd - cd * icd * icd * (k4 + 2 * k5 * r2 + 3 * k6 * r4);
            Gxx = g + 2 * x * x * gp + 2 * p1 * y + 6 * p2 * x;
            Gxy = 2 * x * y * gp + 2 * p1 * x + 2 * p2 * y;
            Gyx = Gxy;
            Gyy = g + 2 * y * y * gp + 6 * p1 * y + 2 * p2 * x;
            const double nd = -cd * icd * icd;
            du[0] = xd; du[1] = 0; du[2] = 1; du[3] = 0;
            dv[0] = 0; dv[1] = yd; dv[2] = 0; dv[3] = 1;
            du[4] = fx * x * r2 * icd; dv[4] = fy * y * r2 * icd;
            du[5] = fx * x * r4 * icd; dv[5] = fy * y * r4 * icd;
            du[6] = fx * a1; dv[6] = fy * a3;
            du[7] = fx * a2; dv[7] = fy * a1;
            du[8] = fx * x * r6 * icd; dv[8] = fy * y * r6 * icd;
            du[9] = fx * x * nd * r2; dv[9] = fy * y * nd * r2;
            du[10] = fx * x * nd * r4; dv[10] = fy * y * nd * r4;
            du[11] = fx * x * nd * r6; dv[11] = fy * y * nd * r6;
        }
    } else {
        alpha = in[4];
        const double k1 = in[5], k2 = in[6], k3 = in[7], k4 = in[8];
        const double r2 = x * x + y * y, r = sqrt(r2);
        const double th = atan(r), t2 = th * th;
        const double thd = th * (1 + t2 * (k1 + t2 * (k2 + t2 * (k3 + t2 * k4))));
        const bool tiny = !(r > 1e-8);
        const double ir = tiny ? 1.0 : 1.0 / r;
        const double sc = tiny ? 1.0 : thd * ir;
        xd = sc * x;
        yd = sc * y;
        if (JAC) {
            const double dthd = 1 + t2 * (3 * k1 + t2 * (5 * k2 + t2 * (7 * k3 + t2 * 9 * k4)));
            const double dsc = tiny ? 0.0 : (dthd / (1 + r2) * r - thd) * ir * ir;  // d scale / d r
            const double cx_ = x * ir, cy_ = y * ir;                                 // d r / d (x, y)
            Gxx = sc + x * dsc * cx_;
            Gxy = x * dsc * cy_;
            Gyx = y * dsc * cx_;
            Gyy = sc + y * dsc * cy_;
            du[0] = xd + alpha * yd; du[1] = 0; du[2] = 1; du[3] = 0;
            dv[0] = 0; dv[1] = yd; dv[2] = 0; dv[3] = 1;
            du[4] = fx * yd; dv[4] = 0;
            const double t3 = t2 * th, t5 = t3 * t2, t7 = t5 * t2, t9 = t7 * t2;
            const double ux = tiny ? 0.0 : x * ir, uy = tiny ? 0.0 : y * ir;
            du[5] = fx * (ux + alpha * uy) * t3; dv[5] = fy * uy * t3;
            du[6] = fx * (ux + alpha * uy) * t5; dv[6] = fy * uy * t5;
            du[7] = fx * (ux + alpha * uy) * t7; dv[7] = fy * uy * t7;
            du[8] = fx * (ux + alpha * uy) * t9; dv[8] = fy * uy * t9;
            du[9] = du[10] = du[11] = 0;
            dv[9] = dv[10] = dv[11] = 0;
        }
    }
    *u = fx * (xd + alpha * yd) + in[2];
    *v = fy * yd + in[3];
    if (JAC) {
        // d(xd, yd)/dY = G * (1/Z) [1 0 -x; 0 1 -y]
        const double Dx0 = Gxx * iz, Dx1 = Gxy * iz, Dx2 = -(Gxx * x + Gxy * y) * iz;
        const double Dy0 = Gyx * iz, Dy1 = Gyy * iz, Dy2 = -(Gyx * x + Gyy * y) * iz;
        P[0][0] = fx * (Dx0 + alpha * Dy0);
        P[0][1] = fx * (Dx1 + alpha * Dy1);
        P[0][2] = fx * (Dx2 + alpha * Dy2);
        P[1][0] = fy * Dy0;
        P[1][1] = fy * Dy1;
        P[1][2] = fy * Dy2;
    }
}

// pixels -> ideal normalised coordinates: cv::undistortPoints (5 fixed-point iterations) / cv::fisheye::undistortPoints
// (Newton on theta, 10 iterations)
__device__ __forceinline__ void undistort_normalized(int model, const double *in, double px, double py, double *ox, double *oy) {
    if (model == 0) {
        const double k1 = in[4], k2 = in[5], p1 = in[6], p2 = in[7], k3 = in[8], k4 = in[9], k5 = in[10], k6 = in[11];
        const double x0 = (px - in[2]) / in[0], y0 = (py - in[3]) / in[1];
        double x = x0, y = y0;
#pragma unroll 1
        for (int it = 0; it < 5; it++) {
            const double r2 = x * x + y * y;
            const double ic = (1 + ((k6 * r2 + k5) * r2 + k4) * r2) / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
            const double dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
            const double dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
            x = (x0 - dx) * ic;
            y = (y0 - dy) * ic;
        }
        *ox = x;
        *oy = y;
        return;
    }
    const double k1 = in[5], k2 = in[6], k3 = in[7], k4 = in[8];
    const double yp = (py - in[3]) / in[1];
    const double xp = (px - in[2]) / in[0] - in[4] * yp;
    double thd = sqrt(xp * xp + yp * yp);
    thd = fmin(fmax(thd, -1.5707963267948966), 1.5707963267948966);
    double th = thd;
#pragma unroll 1
    for (int it = 0; it < 10; it++) {
        const double t2 = th * th;
        const double f = th * (1 + t2 * (k1 + t2 * (k2 + t2 * (k3 + t2 * k4)))) - thd;
        const double df = 1 + t2 * (3 * k1 + t2 * (5 * k2 + t2 * (7 * k3 + t2 * 9 * k4)));
        th -= f / df;
    }
    const double sc = thd > 1e-8 ? tan(th) / thd : 1.0;
    *ox = xp * sc;
    *oy = yp * sc;
}

}  // namespace ecal
