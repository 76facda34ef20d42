// CPU check (g++): the product's analytic residual/Jacobian (eventcalib_amd/csrc/spline_residual.hpp)
// against the oracle's dual-number differentiation of the reference functor.  Test harness only.
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <random>
#include "../../eventcalib_amd/csrc/spline_residual.hpp"

extern "C" double oracle_residual(const double *intr, const double *q4x4, const double *t4x3, const double *basis4,
                                  const double *obs2, const double *lm3, double radius, double *J37, double *J33);
extern "C" double oracle_residual_so3(const double *intr, const double *q4x4, const double *t4x3, const double *basis4,
                                      const double *obs2, const double *lm3, double radius, double *J37, double *J33);
extern "C" double oracle_residual_cam(const double *intr, const double *q4x4, const double *t4x3, const double *basis4,
                                      const double *obs2, const double *lm3, double radius, double *J37, double *J33, int fisheye);
extern "C" double oracle_residual_so3_cam(const double *intr, const double *q4x4, const double *t4x3, const double *basis4,
                                          const double *obs2, const double *lm3, double radius, double *J37, double *J33, int fisheye);
extern "C" uint32_t oracle_find_span(const double *knots, uint32_t n_cp, double u);
extern "C" void oracle_basis(const double *knots, uint32_t span, double u, double *b4);

int main() {
    std::mt19937_64 rng(1234);
    std::uniform_real_distribution<double> U(-1, 1);
    double worst_r = 0, worst_j = 0;
    for (int trial = 0; trial < 20000; trial++) {
        double intr[9] = {359.67525 + 20 * U(rng), 359.67525 + 20 * U(rng), 172.5 + 5 * U(rng), 129.5 + 5 * U(rng),
                          0.35 + 0.05 * U(rng), 0.38 + 0.05 * U(rng), -0.04 + 0.02 * U(rng), -1.16 + 0.1 * U(rng),
                          -4.1 + 0.3 * U(rng)};
        double q[4][4], t[4][3], b[4];
        double base[4] = {0.05 * U(rng), 0.05 * U(rng), 0.7 + 0.05 * U(rng), 0.7 + 0.05 * U(rng)};
        for (int j = 0; j < 4; j++) {
            double n = 0;
            for (int k = 0; k < 4; k++) { q[j][k] = base[k] + 0.02 * U(rng); n += q[j][k] * q[j][k]; }
            n = std::sqrt(n) * (1.0 + 0.01 * U(rng));   // control points are only approximately unit
            for (int k = 0; k < 4; k++) q[j][k] /= n;
            t[j][0] = 19 + 3 * U(rng); t[j][1] = 22 + 3 * U(rng); t[j][2] = -66 + 5 * U(rng);
        }
        // a clamped knot vector with 7 control points and a random parameter
        double knots[11] = {0, 0, 0, 0, 0.21, 0.48, 0.77, 1, 1, 1, 1};
        const double u = 0.5 * (U(rng) + 1.0);
        const uint32_t span = ecal::spline_find_span(knots, 7, u);
        if (span != oracle_find_span(knots, 7, u)) { std::printf("span mismatch\n"); return 1; }
        double ob[4];
        ecal::spline_basis(knots, span, u, b);
        oracle_basis(knots, span, u, ob);
        for (int k = 0; k < 4; k++) if (b[k] != ob[k]) { std::printf("basis mismatch\n"); return 1; }
        double obs[2] = {173 + 150 * U(rng), 130 + 110 * U(rng)};
        double lm[3] = {19 + 18 * U(rng), 22 + 20 * U(rng), 0};
        ecal::ResidualInput in;
        in.ifx = in.ify = 0.0;  // computed inside
        in.u = obs[0]; in.v = obs[1]; in.lmx = lm[0]; in.lmy = lm[1]; in.lmz = lm[2]; in.radius = 1.75;
        for (int k = 0; k < 4; k++) in.b[k] = b[k];
        double J[33], Jo[33];
        const double r = ecal::spline_residual(in, intr, q, t, J);
        const double ro = oracle_residual(intr, &q[0][0], &t[0][0], b, obs, lm, 1.75, nullptr, Jo);
        worst_r = std::fmax(worst_r, std::fabs(r - ro) / (1.0 + std::fabs(ro)));
        double nj = 0;
        for (int i = 0; i < 33; i++) nj = std::fmax(nj, std::fabs(Jo[i]));
        for (int i = 0; i < 33; i++) worst_j = std::fmax(worst_j, std::fabs(J[i] - Jo[i]) / (1e-12 + nj));
    }
    std::printf("max rel residual err %.3e  max rel jacobian err %.3e\n", worst_r, worst_j);
    if (!(worst_r < 1e-12 && worst_j < 1e-10)) return 2;

    // ---- cumulative SO3 spline: analytic tangent Jacobian vs dual numbers through the restated Sophus exp/log
    double so_r = 0, so_j = 0;
    for (int trial = 0; trial < 20000; trial++) {
        double intr[9] = {359.67525 + 20 * U(rng), 359.67525 + 20 * U(rng), 172.5 + 5 * U(rng), 129.5 + 5 * U(rng),
                          0.35 + 0.05 * U(rng), 0.38 + 0.05 * U(rng), -0.04 + 0.02 * U(rng), -1.16 + 0.1 * U(rng),
                          -4.1 + 0.3 * U(rng)};
        double q[4][4], t[4][3], b[4];
        // neighbouring control points differ by rotations from 1e-9 rad (series branches) to ~0.5 rad
        const double spread = trial % 5 == 0 ? 1e-9 : (trial % 5 == 1 ? 1e-4 : (trial % 5 == 2 ? 0.5 : 0.05));
        double base[4] = {0.05 * U(rng), 0.05 * U(rng), 0.7 + 0.05 * U(rng), 0.7 + 0.05 * U(rng)};
        double nb = std::sqrt(base[0] * base[0] + base[1] * base[1] + base[2] * base[2] + base[3] * base[3]);
        for (int k = 0; k < 4; k++) base[k] /= nb;
        for (int j = 0; j < 4; j++) {
            const double w[3] = {spread * U(rng), spread * U(rng), spread * U(rng)};
            ecal::so3_plus(j ? q[j - 1] : base, w, q[j]);
            if (trial % 7 == 3) for (int k = 0; k < 4; k++) q[j][k] = -q[j][k];   // q and -q: same rotation
            t[j][0] = 19 + 3 * U(rng); t[j][1] = 22 + 3 * U(rng); t[j][2] = -66 + 5 * U(rng);
        }
        double knots[11] = {0, 0, 0, 0, 0.21, 0.48, 0.77, 1, 1, 1, 1};
        const double u = 0.5 * (U(rng) + 1.0);
        const uint32_t span = ecal::spline_find_span(knots, 7, u);
        ecal::spline_basis(knots, span, u, b);
        double obs[2] = {173 + 150 * U(rng), 130 + 110 * U(rng)};
        double lm[3] = {19 + 18 * U(rng), 22 + 20 * U(rng), 0};
        ecal::ResidualInput in;
        in.ifx = in.ify = 0.0;  // computed inside
        in.u = obs[0]; in.v = obs[1]; in.lmx = lm[0]; in.lmy = lm[1]; in.lmz = lm[2]; in.radius = 1.75;
        for (int k = 0; k < 4; k++) in.b[k] = b[k];
        double J[33], Jo[33];
        const double r = ecal::spline_residual_so3(in, intr, q, t, J);
        const double ro = oracle_residual_so3(intr, &q[0][0], &t[0][0], b, obs, lm, 1.75, nullptr, Jo);
        so_r = std::fmax(so_r, std::fabs(r - ro) / (1.0 + std::fabs(ro)));
        double nj = 0;
        for (int i = 0; i < 33; i++) nj = std::fmax(nj, std::fabs(Jo[i]));
        for (int i = 0; i < 33; i++) so_j = std::fmax(so_j, std::fabs(J[i] - Jo[i]) / (1e-12 + nj));
    }
    std::printf("SO3: max rel residual err %.3e  max rel jacobian err %.3e\n", so_r, so_j);
    if (!(so_r < 1e-11 && so_j < 1e-8)) return 3;

    // ---- fisheye camera (BASELINE configs[4]: Kannala-Brandt in the inverse form, spline_residual.hpp) with both rotation
    // splines: pixels over the whole sensor, inverse coefficients around the reversion of k = (0.05, -0.01, 0.002, 0)
    double fe_r = 0, fe_j = 0;
    for (int trial = 0; trial < 20000; trial++) {
        double intr[9] = {359.67525 + 20 * U(rng), 359.67525 + 20 * U(rng), 172.5 + 5 * U(rng), 129.5 + 5 * U(rng),
                          -0.05 + 0.02 * U(rng), 0.0175 + 0.01 * U(rng), -0.0075 + 0.005 * U(rng), 0.003 + 0.003 * U(rng),
                          -0.001 + 0.002 * U(rng)};
        double q[4][4], t[4][3], b[4];
        double base[4] = {0.05 * U(rng), 0.05 * U(rng), 0.7 + 0.05 * U(rng), 0.7 + 0.05 * U(rng)};
        double nb = std::sqrt(base[0] * base[0] + base[1] * base[1] + base[2] * base[2] + base[3] * base[3]);
        for (int k = 0; k < 4; k++) base[k] /= nb;
        const bool so3 = trial & 1;
        for (int j = 0; j < 4; j++) {
            if (so3) {
                const double w[3] = {0.05 * U(rng), 0.05 * U(rng), 0.05 * U(rng)};
                ecal::so3_plus(j ? q[j - 1] : base, w, q[j]);
            } else {
                double n = 0;
                for (int k = 0; k < 4; k++) { q[j][k] = base[k] + 0.02 * U(rng); n += q[j][k] * q[j][k]; }
                n = std::sqrt(n) * (1.0 + 0.01 * U(rng));
                for (int k = 0; k < 4; k++) q[j][k] /= n;
            }
            t[j][0] = 19 + 3 * U(rng); t[j][1] = 22 + 3 * U(rng); t[j][2] = -66 + 5 * U(rng);
        }
        double knots[11] = {0, 0, 0, 0, 0.21, 0.48, 0.77, 1, 1, 1, 1};
        const double u = 0.5 * (U(rng) + 1.0);
        const uint32_t span = ecal::spline_find_span(knots, 7, u);
        ecal::spline_basis(knots, span, u, b);
        // every 50th pixel a hair off the principal point: the small-angle branches of c = tan(r poly) / r and of dc/dr2
        const double off = trial % 50 == 0 ? 1e-3 * (1 + trial % 7) : 0.0;
        double obs[2] = {off != 0.0 ? intr[2] + off : 173 + 170 * U(rng), off != 0.0 ? intr[3] - 0.5 * off : 130 + 128 * U(rng)};
        double lm[3] = {19 + 18 * U(rng), 22 + 20 * U(rng), 0};
        ecal::ResidualInput in;
        in.ifx = in.ify = 0.0;
        in.u = obs[0]; in.v = obs[1]; in.lmx = lm[0]; in.lmy = lm[1]; in.lmz = lm[2]; in.radius = 1.75;
        for (int k = 0; k < 4; k++) in.b[k] = b[k];
        double J[33], Jo[33];
        const double r = so3 ? ecal::spline_residual_so3<true>(in, intr, q, t, J) : ecal::spline_residual<true>(in, intr, q, t, J);
        const double ro = (so3 ? oracle_residual_so3_cam : oracle_residual_cam)(intr, &q[0][0], &t[0][0], b, obs, lm, 1.75, nullptr, Jo, 1);
        fe_r = std::fmax(fe_r, std::fabs(r - ro) / (1.0 + std::fabs(ro)));
        double nj = 0;
        for (int i = 0; i < 33; i++) nj = std::fmax(nj, std::fabs(Jo[i]));
        for (int i = 0; i < 33; i++) fe_j = std::fmax(fe_j, std::fabs(J[i] - Jo[i]) / (1e-12 + nj));
    }
    std::printf("fisheye: max rel residual err %.3e  max rel jacobian err %.3e\n", fe_r, fe_j);
    return (fe_r < 1e-11 && fe_j < 1e-7) ? 0 : 4;
}
