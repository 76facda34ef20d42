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
