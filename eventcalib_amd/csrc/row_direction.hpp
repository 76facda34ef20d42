// The direction of a pattern row's line fit (the keyframe gate's input), shared by the keyframe policy (ecal_adaptive.hip) and the
// grid finder's epilogue (ecal_grid.hip): one body, so the two places produce the same bits (both translation units are built
// with -ffp-contract=off).
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <math.h>

namespace ecal {

// direction (B, -A) of the total-least-squares line A x + B y + C = 0 through the row's circle centres: eigenvector of the
// 3x3 Gram matrix of [x y 1] with the smallest eigenvalue (= the right singular vector of EventCalibIni.cpp:46-57), cyclic
// Jacobi as in host/multi_process.hpp; oriented from the first to the last circle.  order: the row's candidate indices (any
// integer type).
template <typename IDX>
__device__ inline void row_direction(const double *xyr, const IDX *order, uint32_t cols, double &dx_out, double &dy_out) {
    double M[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (uint32_t j = 0; j < cols; j++) {
        const double r[3] = {xyr[3 * (size_t) order[j]], xyr[3 * (size_t) order[j] + 1], 1.0};
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) M[a][b] += r[a] * r[b];
    }
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0;
        for (int a = 0; a < 3; a++)
            for (int b = a + 1; b < 3; b++) off += M[a][b] * M[a][b];
        if (off < 1e-300) break;
        for (int a = 0; a < 3; a++)
            for (int b = a + 1; b < 3; b++) {
                if (M[a][b] == 0.0) continue;
                const double th = 0.5 * atan2(2 * M[a][b], M[b][b] - M[a][a]);
                const double c = cos(th), s = sin(th);
                for (int k = 0; k < 3; k++) {
                    const double mka = M[k][a], mkb = M[k][b];
                    M[k][a] = c * mka - s * mkb;
                    M[k][b] = s * mka + c * mkb;
                }
                for (int k = 0; k < 3; k++) {
                    const double mak = M[a][k], mbk = M[b][k];
                    M[a][k] = c * mak - s * mbk;
                    M[b][k] = s * mak + c * mbk;
                }
                for (int k = 0; k < 3; k++) {
                    const double vka = V[k][a], vkb = V[k][b];
                    V[k][a] = c * vka - s * vkb;
                    V[k][b] = s * vka + c * vkb;
                }
            }
    }
    // the smallest diagonal entry, the first of equals (selects instead of a run-time index: the arrays stay in registers)
    int m = 0;
    double dm = M[0][0];
    if (M[1][1] < dm) {
        m = 1;
        dm = M[1][1];
    }
    if (M[2][2] < dm) m = 2;
    double dx = m == 0 ? V[1][0] : m == 1 ? V[1][1] : V[1][2];
    double dy = -(m == 0 ? V[0][0] : m == 1 ? V[0][1] : V[0][2]);
    const double sx = xyr[3 * (size_t) order[cols - 1]] - xyr[3 * (size_t) order[0]];
    const double sy = xyr[3 * (size_t) order[cols - 1] + 1] - xyr[3 * (size_t) order[0] + 1];
    if (dx * sx + dy * sy < 0) {
        dx = -dx;
        dy = -dy;
    }
    dx_out = dx;
    dy_out = dy;
}

}  // namespace ecal
