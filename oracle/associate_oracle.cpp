// TEST INFRASTRUCTURE ONLY — CPU oracle for the event -> residual association.
// Restates event_camera_calib/src/EventCalibSpline.cpp:158-192 (event loop over the spline's time
// range, nearest keyframe in time gated by (5 step)^2) and CirclesEventFrame::findCenter
// (include/opengv2/event_camera_calib/CirclesEventFrame.hpp:50-65: nearest circle centre, accepted if
// |dist - radius| < 5 px).  nanoflann's tie-break between equidistant keyframes / centres is not
// pinned by the reference (third-party, not vendored): restated as "smaller index".
#include <cstdint>
#include <cstring>
#include <cmath>

extern "C" uint64_t oracle_associate(const uint8_t *rec, uint64_t n, const double *kf_time, const double *circles,
                                     uint32_t K, uint32_t n_circ, double t_min, double t_max, double max_dt,
                                     double edge_tol, double *obs, double *time, uint32_t *lm) {
    uint64_t out = 0;
    for (uint64_t i = 0; i < n; i++) {
        double t, x, y;
        std::memcpy(&t, rec + 25 * i, 8);
        std::memcpy(&x, rec + 25 * i + 8, 8);
        std::memcpy(&y, rec + 25 * i + 16, 8);
        if (!(t >= t_min && t <= t_max) || K == 0) continue;
        uint32_t k = 0;
        double bd = (t - kf_time[0]) * (t - kf_time[0]);
        for (uint32_t j = 1; j < K; j++) {
            const double d = (t - kf_time[j]) * (t - kf_time[j]);
            if (d < bd) {
                bd = d;
                k = j;
            }
        }
        if (!(bd < max_dt * max_dt)) continue;
        const double *c = circles + 3 * (size_t) k * n_circ;
        int bi = -1;
        double best = 1.79769313486231570e308;
        for (uint32_t j = 0; j < n_circ; j++) {
            const double dx = x - c[3 * j], dy = y - c[3 * j + 1];
            const double d2 = dx * dx + dy * dy;
            if (d2 < best) {
                best = d2;
                bi = (int) j;
            }
        }
        if (bi < 0) continue;
        if (std::fabs(std::sqrt(best) - c[3 * bi + 2]) < edge_tol) {
            obs[2 * out] = x;
            obs[2 * out + 1] = y;
            time[out] = t;
            lm[out] = (uint32_t) bi;
            out++;
        }
    }
    return out;
}
