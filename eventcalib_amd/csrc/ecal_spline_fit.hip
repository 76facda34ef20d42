// Initial least-squares fit of the pose splines from the keyframe poses, and spline evaluation for the outputs.
//
// Replaces BsplineReal<dim>'s approximating constructor (core/spline/include/opengv2/spline/BsplineReal.hpp:17-100:
// knot placement by averaging, NURBS book (9.68) :87-100; endpoint-interpolating least squares for the interior
// control points :329-449, Eigen::SimplicialLDLT on N^T N) as EventCalibSpline uses it for twb (dim 3) and the
// quaternion coefficients (dim 4) (event_camera_calib/src/EventCalibSpline.cpp:68-91), BsplineSO3::knotSpacing /
// initialGuess (core/spline/src/BsplineSO3.cpp:60-72,198-279: the same fit on the quaternion coefficients), and
// BsplineReal::evaluate(u, 0, ..) (:454-470) for updateMap (EventCalibSpline.cpp:253-317).
// A few hundred keyframes against a few dozen control points: host work.  N^T N of a cubic spline has half
// bandwidth 3, so the system is factorised as a band (LDL^T without pivoting = what SimplicialLDLT computes up to
// its fill-reducing permutation; same solution to rounding).
#include <vector>
#include "ecal_ctx.hpp"
#include "spline_residual.hpp"

using namespace ecal;

extern "C" int ecal_spline_fit(const double *u, const double *data, uint32_t m, uint32_t dim, uint32_t n_cp, double *knots,
                               double *cp) {
    if (!u || !data || !knots || !cp || dim == 0 || n_cp < 4 || m < 2) return ECAL_ERR_INVALID;
    for (uint32_t k = 1; k < m; k++)
        if (!(u[k] >= u[k - 1])) return ECAL_ERR_UNSORTED;
    const uint32_t p = 3;
    // knot vector, NURBS book (9.68)
    for (uint32_t i = 0; i <= p; i++) {
        knots[i] = u[0];
        knots[n_cp + i] = u[m - 1];
    }
    const double d = m / double(n_cp - p);
    for (uint32_t j = 1; j + p + 1 <= n_cp; j++) {
        const int i = (int) floor(j * d);
        const double alpha = j * d - i;
        if (i < 1 || (uint32_t) i >= m) return ECAL_ERR_INVALID;  // fewer samples than spans
        knots[p + j] = (1 - alpha) * u[i - 1] + alpha * u[i];
    }
    for (uint32_t c = 0; c < dim; c++) {
        cp[c] = data[c];
        cp[(size_t) (n_cp - 1) * dim + c] = data[(size_t) (m - 1) * dim + c];
    }
    const uint32_t ni = n_cp - 2;  // interior control points
    if (ni == 0) return ECAL_OK;
    // band storage of A = Nc^T Nc: A[i][i + o], o = 0..3
    std::vector<double> A((size_t) ni * 4, 0.0), B((size_t) ni * dim, 0.0);
    for (uint32_t k = 1; k + 1 < m; k++) {  // rows 0 and m-1 are interpolated, not fitted (:401)
        const uint32_t span = spline_find_span(knots, n_cp, u[k]);
        double b[4];
        spline_basis(knots, span, u[k], b);
        double n_first = 0, n_last = 0;
        for (int j = 0; j < 4; j++) {
            const uint32_t col = span - p + j;
            if (col == 0) n_first = b[j];
            if (col == n_cp - 1) n_last = b[j];
        }
        for (int j = 0; j < 4; j++) {
            const uint32_t col = span - p + j;
            if (col == 0 || col == n_cp - 1 || b[j] == 0) continue;
            for (uint32_t c = 0; c < dim; c++)
                B[(size_t) (col - 1) * dim + c] += b[j] * (data[(size_t) k * dim + c] - n_first * data[c] -
                                                           n_last * data[(size_t) (m - 1) * dim + c]);
            for (int j2 = j; j2 < 4; j2++) {
                const uint32_t col2 = span - p + j2;
                if (col2 == 0 || col2 == n_cp - 1) continue;
                A[(size_t) (col - 1) * 4 + (col2 - col)] += b[j] * b[j2];
            }
        }
    }
    // banded Cholesky A = L L^T (lower band kept in L[i][o] = L(i, i - o))
    std::vector<double> L((size_t) ni * 4, 0.0);
    for (uint32_t i = 0; i < ni; i++) {
        for (uint32_t o = (i < 3 ? i : 3) + 1; o-- > 0;) {  // columns j = i - o ... i
            const uint32_t j = i - o;
            double s = A[(size_t) j * 4 + o];  // A(i, j) = A(j, i), stored at row j offset o
            for (uint32_t k = (i < 3 ? 0 : i - 3); k < j; k++) {
                if (j - k > 3) continue;
                s -= L[(size_t) i * 4 + (i - k)] * L[(size_t) j * 4 + (j - k)];
            }
            if (o == 0) {
                if (!(s > 0)) return ECAL_ERR_INVALID;  // a control point no sample supports
                L[(size_t) i * 4] = sqrt(s);
            } else {
                L[(size_t) i * 4 + o] = s / L[(size_t) j * 4];
            }
        }
    }
    std::vector<double> x(ni);
    for (uint32_t c = 0; c < dim; c++) {
        for (uint32_t i = 0; i < ni; i++) {
            double s = B[(size_t) i * dim + c];
            for (uint32_t o = 1; o <= 3 && o <= i; o++) s -= L[(size_t) i * 4 + o] * x[i - o];
            x[i] = s / L[(size_t) i * 4];
        }
        for (uint32_t i = ni; i-- > 0;) {
            double s = x[i];
            for (uint32_t o = 1; o <= 3 && i + o < ni; o++) s -= L[(size_t) (i + o) * 4 + o] * x[i + o];
            x[i] = s / L[(size_t) i * 4];
        }
        for (uint32_t i = 0; i < ni; i++) cp[(size_t) (i + 1) * dim + c] = x[i];
    }
    return ECAL_OK;
}

extern "C" int ecal_spline_eval(const double *knots, const double *cp, uint32_t n_cp, uint32_t dim, const double *u, uint32_t m,
                                double *out) {
    if (!knots || !cp || !u || !out || n_cp < 4 || dim == 0) return ECAL_ERR_INVALID;
    for (uint32_t k = 0; k < m; k++) {
        if (!(u[k] >= knots[0] && u[k] <= knots[n_cp])) return ECAL_ERR_RANGE;  // "spline can only evaluate inside the bound"
        const uint32_t span = spline_find_span(knots, n_cp, u[k]);
        double b[4];
        spline_basis(knots, span, u[k], b);
        for (uint32_t c = 0; c < dim; c++) {
            double s = 0;
            for (int j = 0; j < 4; j++) s += b[j] * cp[(size_t) (span - 3 + j) * dim + c];
            out[(size_t) k * dim + c] = s;
        }
    }
    return ECAL_OK;
}

// ------------------------------------------------------------------------------------------------
// BsplineSO3::optimizeCP (core/spline/src/BsplineSO3.cpp:285-341): the control points of the cumulative cubic SO3 spline
// refined on the group — residual of sample i (P3ApproximationError, BsplineSO3.hpp:121-153)
//     r_i = log( S_i^-1  cp_0 exp(b1 log(cp_0^-1 cp_1)) exp(b2 log(cp_1^-1 cp_2)) exp(b3 log(cp_2^-1 cp_3)) )   in R^3,
// b_j = the cumulative basis at u_i, cp_0..3 = the control points of u_i's span; unknowns = rotation-vector steps
// cp <- cp exp(delta) (LocalParameterizationSO3, BsplineSO3.hpp:190-222); first and last control point constant (:294-295);
// no loss function; function / gradient tolerance 1e-10 (:330-331), SPARSE_NORMAL_CHOLESKY, Ceres' default 50 iterations.
// Ceres' autodiff is replaced by the closed-form derivative (the chain through the cumulative factors that
// spline_residual_so3 uses, here for the three rows of Jr^-1(r)), its trust-region loop by the same restatement as
// ecal_solver_solve; J^T J couples a control point with the next three: a band of half-width 12, factorised on the host
// (a few hundred keyframes, a few dozen control points).
// ------------------------------------------------------------------------------------------------
namespace {

struct So3Sample {
    uint32_t span;
    double beta[3];
    double sinv[4];
};

// residual r[3] and, if J != nullptr, J[3][12]: row k, columns 3 j + c = d r_k / d delta_c of control point j of the span
void so3_fit_residual(const So3Sample &s, const double (*q)[4], double r[3], double (*J)[12]) {
    double d[3][3], bd[3][3], Q[4] = {q[0][0], q[0][1], q[0][2], q[0][3]};
    for (int j = 1; j <= 3; j++) {
        const double inv[4] = {-q[j - 1][0], -q[j - 1][1], -q[j - 1][2], q[j - 1][3]};
        double rel[4], e[4], nq[4];
        quat_mul(inv, q[j], rel);
        so3_log(rel, d[j - 1]);
        for (int k = 0; k < 3; k++) bd[j - 1][k] = s.beta[j - 1] * d[j - 1][k];
        so3_exp(bd[j - 1], e);
        quat_mul(Q, e, nq);
        for (int k = 0; k < 4; k++) Q[k] = nq[k];
    }
    double E[4];
    quat_mul(s.sinv, Q, E);
    so3_log(E, r);
    if (!J) return;
    // X <- X exp(w):  r(w) = log(exp(r) exp(w))  =>  d r / d w = Jr^-1(r); row k of it is Jl^-1(r) e_k (Jr^-1 = Jl^-T ... as
    // a covector: v_k = Jr^-1(r)^T e_k = Jl^-1(r) e_k), pushed back through the factors as in spline_residual_so3
    for (int k = 0; k < 3; k++) {
        double ek[3] = {0, 0, 0}, v[3], m[3], a[3], n[3];
        ek[k] = 1.0;
        so3_jinv(r, +1.0, ek, v);
        double p_next[3] = {0, 0, 0};
        for (int j = 3; j >= 1; j--) {
            so3_jl(bd[j - 1], v, m);
            for (int c = 0; c < 3; c++) m[c] *= s.beta[j - 1];
            so3_jinv(d[j - 1], +1.0, m, a);
            so3_jinv(d[j - 1], -1.0, m, n);
            for (int c = 0; c < 3; c++) J[k][3 * j + c] = a[c] - (j < 3 ? p_next[c] : 0.0);
            for (int c = 0; c < 3; c++) p_next[c] = n[c];
            so3_rotate(bd[j - 1], v, a);
            for (int c = 0; c < 3; c++) v[c] = a[c];
        }
        for (int c = 0; c < 3; c++) J[k][c] = v[c] - p_next[c];
    }
}

}  // namespace

extern "C" int ecal_spline_so3_refine(const double *knots, uint32_t n_cp, double *cp_quat, const double *sample_quat, const double *u,
                                      uint32_t m, int max_iterations, double *initial_cost, double *final_cost, int *iterations) {
    if (!knots || !cp_quat || !sample_quat || !u || n_cp < 4 || m < 1) return ECAL_ERR_INVALID;
    if (max_iterations <= 0) max_iterations = 50;
    std::vector<So3Sample> S(m);
    for (uint32_t i = 0; i < m; i++) {
        if (!(u[i] >= knots[0] && u[i] <= knots[n_cp])) return ECAL_ERR_RANGE;
        const uint32_t span = spline_find_span(knots, n_cp, u[i]);
        double b[4];
        spline_basis(knots, span, u[i], b);
        S[i].span = span;
        S[i].beta[2] = b[3];                 // derBasisFuns(u, span, 0): cumulative sums (BsplineSO3.cpp:88-94)
        S[i].beta[1] = S[i].beta[2] + b[2];
        S[i].beta[0] = S[i].beta[1] + b[1];
        const double *q = sample_quat + 4 * (size_t) i;
        const double nq = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        if (!(nq > 0)) return ECAL_ERR_INVALID;
        S[i].sinv[0] = -q[0] / nq;
        S[i].sinv[1] = -q[1] / nq;
        S[i].sinv[2] = -q[2] / nq;
        S[i].sinv[3] = q[3] / nq;
    }
    std::vector<double> x(cp_quat, cp_quat + 4 * (size_t) n_cp), xc(4 * (size_t) n_cp);
    for (uint32_t c = 0; c < n_cp; c++) {   // unit quaternions (Sophus::SO3d::setQuaternion normalises)
        double *q = &x[4 * (size_t) c];
        const double nq = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        if (!(nq > 0)) return ECAL_ERR_INVALID;
        for (int k = 0; k < 4; k++) q[k] /= nq;
    }
    const uint32_t ni = n_cp - 2;           // free control points 1 .. n_cp - 2
    const size_t nu = 3 * (size_t) ni;
    constexpr int HB = 12;                  // band: unknown a couples with a - 11 .. a + 11
    std::vector<double> H(nu * HB), g(nu), L(nu * HB), delta(nu), scale(nu, 1.0);
    auto cost_of = [&](const std::vector<double> &xx) {
        double c = 0;
        for (uint32_t i = 0; i < m; i++) {
            double r[3];
            so3_fit_residual(S[i], reinterpret_cast<const double(*)[4]>(&xx[4 * (size_t) (S[i].span - 3)]), r, nullptr);
            c += 0.5 * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        }
        return c;
    };
    auto normal = [&](const std::vector<double> &xx) {   // H (lower band: H[a][k] = H(a, a - k)), g; returns the cost
        std::fill(H.begin(), H.end(), 0.0);
        std::fill(g.begin(), g.end(), 0.0);
        double c = 0;
        for (uint32_t i = 0; i < m; i++) {
            double r[3], J[3][12];
            const uint32_t c0 = S[i].span - 3;
            so3_fit_residual(S[i], reinterpret_cast<const double(*)[4]>(&xx[4 * (size_t) c0]), r, J);
            c += 0.5 * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
            for (int a = 0; a < 12; a++) {
                const uint32_t ca = c0 + a / 3;
                if (ca == 0 || ca == n_cp - 1) continue;   // constant blocks
                const size_t ia = 3 * (size_t) (ca - 1) + a % 3;
                g[ia] += J[0][a] * r[0] + J[1][a] * r[1] + J[2][a] * r[2];
                for (int b = 0; b <= a; b++) {
                    const uint32_t cb = c0 + b / 3;
                    if (cb == 0 || cb == n_cp - 1) continue;
                    const size_t ib = 3 * (size_t) (cb - 1) + b % 3;
                    H[ia * HB + (ia - ib)] += J[0][a] * J[0][b] + J[1][a] * J[1][b] + J[2][a] * J[2][b];
                }
            }
        }
        return c;
    };
    auto solve = [&](double radius) -> bool {   // (S H S + D) y = -S g, delta = S y; false if not positive definite
        for (size_t a = 0; a < nu; a++) {
            for (int k = 0; k < HB; k++) L[a * HB + k] = (size_t) k <= a ? H[a * HB + k] * scale[a] * scale[a - k] : 0.0;
            const double h = L[a * HB];
            L[a * HB] += std::min(std::max(h, 1e-6), 1e32) / radius;
        }
        for (size_t a = 0; a < nu; a++) {
            const int kmax = (int) std::min<size_t>(HB - 1, a);
            for (int k = kmax; k >= 0; k--) {   // column b = a - k
                const size_t b = a - k;
                double v = L[a * HB + k];
                for (int t = k + 1; t <= kmax; t++) {       // common columns c = a - t < b, with b - c = t - k < HB
                    if (t - k >= HB) break;
                    v -= L[a * HB + t] * L[b * HB + (t - k)];
                }
                if (k == 0) {
                    if (!(v > 0.0)) return false;
                    L[a * HB] = sqrt(v);
                } else {
                    L[a * HB + k] = v / L[b * HB];
                }
            }
        }
        for (size_t a = 0; a < nu; a++) {
            double v = -g[a] * scale[a];
            const int kmax = (int) std::min<size_t>(HB - 1, a);
            for (int k = 1; k <= kmax; k++) v -= L[a * HB + k] * delta[a - k];
            delta[a] = v / L[a * HB];
        }
        for (size_t a = nu; a-- > 0;) {
            double v = delta[a];
            for (int k = 1; k < HB && a + k < nu; k++) v -= L[(a + k) * HB + k] * delta[a + k];
            delta[a] = v / L[a * HB];
        }
        for (size_t a = 0; a < nu; a++) delta[a] *= scale[a];
        return true;
    };
    double cost = normal(x);
    if (initial_cost) *initial_cost = cost;
    int it = 0;
    if (nu > 0) {
        for (size_t a = 0; a < nu; a++) scale[a] = 1.0 / (1.0 + sqrt(H[a * HB]));   // Jacobi scaling, fixed at the start (Ceres)
        auto gmax = [&]() {
            double mx = 0;
            for (size_t a = 0; a < nu; a++) mx = std::max(mx, fabs(g[a]));
            return mx;
        };
        double radius = 1e4, decrease = 2.0;
        bool done = gmax() <= 1e-10;
        while (!done && it < max_iterations) {
            it++;
            bool ok = solve(radius);
            double model = 0;
            if (ok) {   // model cost change -g^T d - d^T H d / 2
                double gd = 0, dHd = 0;
                for (size_t a = 0; a < nu; a++) {
                    gd += g[a] * delta[a];
                    double row = H[a * HB] * delta[a];
                    const int kmax = (int) std::min<size_t>(HB - 1, a);
                    for (int k = 1; k <= kmax; k++) row += 2.0 * H[a * HB + k] * delta[a - k];
                    dHd += delta[a] * row;
                }
                model = -gd - 0.5 * dHd;
                ok = model > 0.0;
            }
            if (!ok) {
                radius /= decrease;
                decrease *= 2.0;
                continue;
            }
            xc = x;
            double step2 = 0, x2 = 0;
            for (uint32_t c = 1; c + 1 < n_cp; c++) so3_plus(&x[4 * (size_t) c], &delta[3 * (size_t) (c - 1)], &xc[4 * (size_t) c]);
            for (size_t a = 0; a < nu; a++) step2 += delta[a] * delta[a];
            for (size_t a = 0; a < x.size(); a++) x2 += x[a] * x[a];
            const double new_cost = cost_of(xc);
            const double rel = (cost - new_cost) / model;
            if (rel > 1e-3) {
                const double change = cost - new_cost, prev = cost;
                x.swap(xc);
                cost = normal(x);
                const double t = 2.0 * rel - 1.0;
                radius = std::min(1e16, radius / std::max(1.0 / 3.0, 1.0 - t * t * t));
                decrease = 2.0;
                if (gmax() <= 1e-10 || fabs(change) <= 1e-10 * prev) done = true;
            } else {
                radius /= decrease;
                decrease *= 2.0;
            }
            if (!done && sqrt(step2) <= 1e-8 * (sqrt(x2) + 1e-8)) done = true;
        }
    }
    memcpy(cp_quat, x.data(), x.size() * sizeof(double));
    if (final_cost) *final_cost = cost;
    if (iterations) *iterations = it;
    return ECAL_OK;
}
