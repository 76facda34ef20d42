// The linear solve of one Levenberg-Marquardt step ON THE DEVICE: (S A S + D) y = -S g for the block-banded arrow system
// of the continuous-time calibration (what Ceres' SPARSE_NORMAL_CHOLESKY does for the reference,
// event_camera_calib/src/EventCalibSpline.cpp:241).  Unknown order [control point c: d_rot 3, d_trans 3 | ... | intrinsics 9]:
// a symmetric band of 4 blocks of 6 (a control point couples with its three successors: cubic splines) with a dense 9-wide
// border.  A banded Cholesky is a chain of 6 n_cp dependent column steps — 12 000 for the benchmark's 2000 control points,
// ~1.3 ms on a host core and no faster on one wavefront.  It is made parallel by substructuring: the block columns are cut
// into P partitions of K interior columns separated by 3 columns each (a separator as wide as the band: interiors of
// different partitions do not couple); with the unknowns ordered [interiors | separators | intrinsics]
//   1. arrow_interior_kernel   P wavefronts: every interior is factorised on its own, its couplings to the separator on
//                              its left, the one on its right, the intrinsics and the right-hand side carried as a 46-wide
//                              border Z = L^-1 [B_left B_right B_intr rhs]; G_p = Z^T Z is its Schur contribution;
//   2. arrow_reduced_kernel    1 wavefront: the separators' system (18-wide blocks, block tridiagonal, minus the G_p) with
//                              the intrinsics as border is factorised the same way; the 10 x 10 sums [Zb z]^T [Zb z] come out;
//   3. arrow_corner_kernel     the 9 x 9 intrinsics system, then the separators' back substitution;
//   4. arrow_backsub_kernel    P wavefronts: the interiors' back substitution, the step d = S y.
// Both factorisations are the same routine (band_border_factor): right-looking scalar columns over a sliding window of the
// band's rows kept in LDS.  Checked against the host solve (solve_arrow, ecal_solver.hip) in tests/test_gpu_solver.py.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ecal {

constexpr int AR_BW = 24;                         // scalar band incl. the diagonal (4 blocks of 6)
constexpr int AR_SEP = 18;                        // separator: 3 block columns
constexpr int AR_NB = 2 * AR_SEP + 10;            // interior border: left separator, right separator, intrinsics, rhs
constexpr int AR_RBW = 2 * AR_SEP;                // reduced band incl. the diagonal (2 blocks of 18)
constexpr int AR_RNB = 10;                        // reduced border: intrinsics, rhs
constexpr size_t AR_HEAD = 91, AR_PER_CP = 204;   // layout of the normal-equation buffer (ecal_solver.hip)

struct ArrowPlan {
    uint32_t n_cp, K, P;   // P partitions: interiors of K block columns (the last one takes the rest), P - 1 separators
    __host__ __device__ uint32_t int_lo(uint32_t p) const { return p * (K + 3); }                      // block columns
    __host__ __device__ uint32_t int_hi(uint32_t p) const { return p + 1 == P ? n_cp : p * (K + 3) + K; }
    __host__ __device__ uint32_t n_red() const { return AR_SEP * (P - 1); }
};

struct ArrowLm {
    double radius, min_diag, max_diag;
};

// A(row, col), row >= col, of the control-point part, straight out of the normal-equation buffer (upper blocks stored)
__device__ __forceinline__ double ar_band(const double *__restrict__ acc, uint32_t row, uint32_t col) {
    const uint32_t rb = row / 6u, cb = col / 6u, d = rb - cb;
    if (d > 3u) return 0.0;
    return acc[AR_HEAD + AR_PER_CP * (size_t) cb + 60 + 36 * d + 6 * (col % 6u) + (row % 6u)];
}
__device__ __forceinline__ double ar_border(const double *__restrict__ acc, uint32_t row, int j) {
    return acc[AR_HEAD + AR_PER_CP * (size_t) (row / 6u) + 6 + 9 * (row % 6u) + j];
}
__device__ __forceinline__ double ar_gc(const double *__restrict__ acc, uint32_t row) {
    return acc[AR_HEAD + AR_PER_CP * (size_t) (row / 6u) + (row % 6u)];
}
__device__ __forceinline__ double ar_corner(const double *__restrict__ acc, int i, int j) {
    return i <= j ? acc[10 + 9 * i + j] : acc[10 + 9 * j + i];
}
__device__ __forceinline__ double ar_lm_diag(double h_scaled, const ArrowLm lm) {
    return fmin(fmax(h_scaled, lm.min_diag), lm.max_diag) / lm.radius;
}

__device__ __forceinline__ void ar_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// sum of a double over the 64 lanes, valid in lane 63 (DPP row shifts / broadcasts on the two halves: no LDS round trips;
// __shfl_xor costs two ds_bpermute per step)
template <int ctrl, int row_mask, int bank_mask>
__device__ __forceinline__ double ar_dpp_shift_add(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int) (b & 0xFFFFFFFFll), ctrl, row_mask, bank_mask, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int) (b >> 32), ctrl, row_mask, bank_mask, false);
    return v + __longlong_as_double(((long long) hi << 32) | (unsigned int) lo);   // (lanes without a source add +0.0)
}
__device__ __forceinline__ double ar_wave_sum_to_last(double v) {
    v = ar_dpp_shift_add<0x111, 0xF, 0xF>(v);   // row_shr:1
    v = ar_dpp_shift_add<0x112, 0xF, 0xF>(v);   // row_shr:2  (now 4-lane sums at lanes 3, 7, ... : each lane holds itself + 3 before it)
    v = ar_dpp_shift_add<0x114, 0xF, 0xE>(v);   // row_shr:4
    v = ar_dpp_shift_add<0x118, 0xF, 0xC>(v);   // row_shr:8  -> lane 15 of every row: the row's sum
    v = ar_dpp_shift_add<0x142, 0xA, 0xF>(v);   // row_bcast:15
    v = ar_dpp_shift_add<0x143, 0xC, 0xF>(v);   // row_bcast:31 -> lane 63: everything
    return v;
}

#ifdef ECAL_PHASE_PROF
__device__ unsigned long long g_ar_cycles[8];
#define AR_MARK(i)                                                                       \
    do {                                                                                 \
        if (lane == 0 && blockIdx.x == 0) {                                              \
            const unsigned long long now__ = __builtin_amdgcn_s_memtime();               \
            atomicAdd(&g_ar_cycles[i], now__ - ar_t__);                                  \
            ar_t__ = now__;                                                              \
        }                                                                                \
    } while (0)
#else
#define AR_MARK(i)
#endif

// One wavefront: Cholesky of a banded matrix of n rows (n a multiple of BLK; band BWT incl. the diagonal, BWT a multiple of
// BLK, zero outside blocks of BLK: a column of block column J reaches down to the last row of block J + BWT / BLK - 1) with
// NB border columns carried along:   L L^T = band,  Z = L^-1 border.
// fetch(i, c) returns entry c of the stored row [BWT + NB] of matrix row i (band offsets 0 .. BWT-1: A(i, i-k); then the
// border) — a pure function, so that a block of rows is requested with ALL its global loads in flight at once.  Finished
// rows go to L_out [n][BWT] / Z_out [n][NB]; with WITH_G, G_out (NB x NB, full storage) receives Z^T Z.
// BLOCK steps: a chain of scalar column steps costs a few hundred instructions of overhead per column; here a block column
// (BLK scalar columns) is one step —
//   1. the diagonal block and its BLK pivot rows are finished by scalar steps confined to those rows;
//   2. every row below (RB = BWT - BLK of them, one lane each) gets its BLK entries of the block column by forward
//      substitution against the diagonal block: the panel pn[RB][BLK];
//   3. the trailing update  row_i -= pn[i] . (pn[c] | Z_pivot[.][c])  with ONE LANE PER COLUMN of the window row (RB band
//      columns + NB border columns: 64 lanes for the interiors): the lane's BLK factors stay in registers, pn[i] is a
//      broadcast read, one LDS read-modify-write per row and lane.
// win: (BWT + BLK) x (BWT + NB) doubles of LDS (a freed block of slots is refilled with one batch of loads), pn: RB x BLK.
// Returns false when a pivot is not positive.
template <int BWT, int NB, int BLK, bool WITH_G, class Loader>
__device__ bool band_border_factor(uint32_t n, Loader fetch, double *__restrict__ L_out, double *__restrict__ Z_out,
                                   double *__restrict__ G_out, double *win, double *colv, double *pn) {
    constexpr int W = BWT + NB, RB = BWT - BLK;
    constexpr uint32_t WR = (uint32_t) (BWT + BLK);
    static_assert(RB + NB <= 64 && BWT % BLK == 0, "one lane per column of the trailing update");
    constexpr int NG = NB * (NB + 1) / 2;
    static_assert(!WITH_G || NG <= 64, "G: one entry per lane");
    const uint32_t lane = threadIdx.x & 63u;
    double g = 0.0;
    uint32_t ga = 0, gb = 0;
    if (WITH_G) {
        uint32_t e = lane < (uint32_t) NG ? lane : 0u;
        while (e >= (uint32_t) NB - ga) {
            e -= (uint32_t) NB - ga;
            ga++;
        }
        gb = ga + e;
    }
    constexpr int CPL = (W + 63) / 64;   // stored columns per lane
    double tmp[BLK][CPL];                // a block of rows on its way from global memory to the window
    auto fetch_block = [&](uint32_t i0) {   // rows i0 .. i0 + BLK - 1: all loads issued together
#pragma unroll
        for (int q = 0; q < BLK; q++)
#pragma unroll
            for (int m = 0; m < CPL; m++) {
                const uint32_t c = lane + 64u * m;
                tmp[q][m] = (i0 + q < n && c < (uint32_t) W) ? fetch(i0 + (uint32_t) q, c) : 0.0;
            }
    };
    auto store_block = [&](uint32_t i0, uint32_t slot) {
#pragma unroll
        for (int q = 0; q < BLK; q++) {
            uint32_t sl = slot + (uint32_t) q;
            if (sl >= WR) sl -= WR;
#pragma unroll
            for (int m = 0; m < CPL; m++) {
                const uint32_t c = lane + 64u * m;
                if (i0 + q < n && c < (uint32_t) W) win[(size_t) sl * W + c] = tmp[q][m];
            }
        }
    };
    for (uint32_t i = 0; i < WR && i < n; i += (uint32_t) BLK) {
        fetch_block(i);
        store_block(i, i % WR);
    }
    ar_wave_sync();
    bool ok = true;
#ifdef ECAL_PHASE_PROF
    unsigned long long ar_t__ = __builtin_amdgcn_s_memtime();
#endif
    uint32_t slot0 = 0;   // window slot of row j0
    for (uint32_t j0 = 0; j0 < n && ok; j0 += (uint32_t) BLK) {
        auto rowp = [&](uint32_t r) -> double * {   // window row of matrix row j0 + r (r < WR)
            uint32_t sl = slot0 + r;
            if (sl >= WR) sl -= WR;
            return win + (size_t) sl * W;
        };
        const bool more = j0 + WR < n;
        AR_MARK(0);
        if (more) fetch_block(j0 + WR);   // the rows that will take this block's slots: requested now, stored at the end
        AR_MARK(1);
        // 1. the pivot rows
        for (uint32_t q = 0; q < (uint32_t) BLK; q++) {
            double *rowj = rowp(q);
            const double piv = rowj[0];
            if (!(piv > 0.0)) {
                ok = false;
                break;
            }
            const double d = sqrt(piv), inv = 1.0 / d;
            const uint32_t rin = (uint32_t) BLK - 1u - q;   // rows of the block below the pivot
            ar_wave_sync();
            if (lane == 0) {
                rowj[0] = d;
                colv[BWT + q] = inv;   // (kept for the panel solve: multiplications instead of divisions)
            }
            if (lane < (uint32_t) NB) rowj[BWT + lane] *= inv;
            if (lane >= 1u && lane <= rin) {
                double *rr = rowp(q + lane);
                const double v = rr[lane] * inv;
                rr[lane] = v;
                colv[lane] = v;
            }
            ar_wave_sync();
            // lane t < BLK - 1: band column c = t + 1 of the block; lanes BLK .. BLK + NB - 1: the border columns
            if (lane < rin) {
                const double pc = colv[lane + 1u];
                for (uint32_t r = lane + 1u; r <= rin; r++) rowp(q + r)[r - lane - 1u] -= colv[r] * pc;
            } else if (lane >= (uint32_t) BLK && lane < (uint32_t) (BLK + NB)) {
                const uint32_t bc = (uint32_t) BWT + lane - (uint32_t) BLK;
                const double pz = rowj[bc];
                for (uint32_t r = 1; r <= rin; r++) rowp(q + r)[bc] -= colv[r] * pz;
            }
            ar_wave_sync();
        }
        if (!ok) break;
        AR_MARK(2);
        const uint32_t left = n - j0 - (uint32_t) BLK, nr = left < (uint32_t) RB ? left : (uint32_t) RB;   // rows below
        // 2. the panel: row j0 + BLK + ri solves  pn[ri] Ld^T = A(row, block)
        if (lane < nr) {
            double *rr = rowp((uint32_t) BLK + lane);
            double x[BLK];
#pragma unroll
            for (int q = 0; q < BLK; q++) {
                const double *pq = rowp((uint32_t) q);
                double v = rr[(uint32_t) BLK + lane - (uint32_t) q];
#pragma unroll
                for (int q2 = 0; q2 < q; q2++) v -= x[q2] * pq[q - q2];
                x[q] = v * colv[BWT + q];
                rr[(uint32_t) BLK + lane - (uint32_t) q] = x[q];
                pn[lane * BLK + q] = x[q];
            }
        }
        ar_wave_sync();
        AR_MARK(3);
        // 3. trailing update, one lane per column
        if (nr) {
            double mine[BLK];
            const bool is_band = lane < (uint32_t) RB, is_border = lane >= (uint32_t) RB && lane < (uint32_t) (RB + NB);
#pragma unroll
            for (int q = 0; q < BLK; q++) {
                mine[q] = 0.0;
                if (is_band && lane < nr) mine[q] = pn[lane * BLK + q];
                else if (is_border) mine[q] = rowp((uint32_t) q)[(uint32_t) BWT + lane - (uint32_t) RB];
            }
            // (all the window reads first, then the writes: nothing here aliases, but the compiler cannot know that)
            double upd[RB];
#pragma unroll
            for (int ri = 0; ri < RB; ri++) {
                double acc = 0.0;
#pragma unroll
                for (int q = 0; q < BLK; q++) acc += pn[ri * BLK + q] * mine[q];
                const double *rr = rowp((uint32_t) BLK + (uint32_t) ri);
                double old = 0.0;
                if ((uint32_t) ri < nr) {
                    if (is_band) {
                        if (lane <= (uint32_t) ri) old = rr[(uint32_t) ri - lane];
                    } else if (is_border) {
                        old = rr[(uint32_t) BWT + lane - (uint32_t) RB];
                    }
                }
                upd[ri] = old - acc;
            }
#pragma unroll
            for (int ri = 0; ri < RB; ri++) {
                double *rr = rowp((uint32_t) BLK + (uint32_t) ri);
                if ((uint32_t) ri < nr) {
                    if (is_band) {
                        if (lane <= (uint32_t) ri) rr[(uint32_t) ri - lane] = upd[ri];
                    } else if (is_border) {
                        rr[(uint32_t) BWT + lane - (uint32_t) RB] = upd[ri];
                    }
                }
            }
        }
        AR_MARK(4);
        // the BLK pivot rows are final: out they go
        for (uint32_t q = 0; q < (uint32_t) BLK; q++) {
            const double *rowj = rowp(q);
            for (uint32_t c = lane; c < (uint32_t) W; c += 64u) {
                if (c < (uint32_t) BWT) L_out[(size_t) (j0 + q) * BWT + c] = rowj[c];
                else Z_out[(size_t) (j0 + q) * NB + (c - BWT)] = rowj[c];
            }
            if (WITH_G) g += rowj[BWT + ga] * rowj[BWT + gb];
        }
        ar_wave_sync();
        AR_MARK(5);
        if (more) store_block(j0 + WR, slot0);
        ar_wave_sync();
        AR_MARK(6);
        slot0 += (uint32_t) BLK;
        if (slot0 >= WR) slot0 -= WR;
    }
    if (WITH_G && lane < (uint32_t) NG) {
        G_out[ga * NB + gb] = g;
        G_out[gb * NB + ga] = g;
    }
    return ok;
}

// G_p = Z^T Z of every interior (46 x 46, full storage): rows staged through LDS, a few entries per thread
__global__ __launch_bounds__(256) void arrow_gram_kernel(ArrowPlan plan, const double *__restrict__ Zfac, double *__restrict__ Gp) {
    constexpr int TR = 32, NG = AR_NB * (AR_NB + 1) / 2, EPT = (NG + 255) / 256;
    __shared__ double zt[TR * AR_NB];
    const uint32_t p = blockIdx.x, tid = threadIdx.x;
    const uint32_t lo = 6u * plan.int_lo(p), n = 6u * plan.int_hi(p) - lo;
    uint32_t ea[EPT], eb[EPT];
    double acc[EPT];
#pragma unroll
    for (int m = 0; m < EPT; m++) {
        acc[m] = 0.0;
        uint32_t e = tid + 256u * m, a = 0;
        if (e >= (uint32_t) NG) e = 0;
        while (e >= (uint32_t) AR_NB - a) {
            e -= (uint32_t) AR_NB - a;
            a++;
        }
        ea[m] = a;
        eb[m] = a + e;
    }
    for (uint32_t r0 = 0; r0 < n; r0 += TR) {
        const uint32_t nr = n - r0 < (uint32_t) TR ? n - r0 : (uint32_t) TR;
        for (uint32_t k = tid; k < nr * AR_NB; k += 256u) zt[k] = Zfac[(size_t) (lo + r0) * AR_NB + k];
        __syncthreads();
        for (uint32_t r = 0; r < nr; r++) {
#pragma unroll
            for (int m = 0; m < EPT; m++) acc[m] += zt[r * AR_NB + ea[m]] * zt[r * AR_NB + eb[m]];
        }
        __syncthreads();
    }
#pragma unroll
    for (int m = 0; m < EPT; m++) {
        if (tid + 256u * m < (uint32_t) NG) {
            Gp[(size_t) p * AR_NB * AR_NB + ea[m] * AR_NB + eb[m]] = acc[m];
            Gp[(size_t) p * AR_NB * AR_NB + eb[m] * AR_NB + ea[m]] = acc[m];
        }
    }
}

// Jacobi column scaling from the initial normal matrix, as Ceres computes it once: 1 / (1 + sqrt(diagonal))
__global__ void arrow_scale_kernel(const double *__restrict__ acc, uint32_t n_cp, int jacobi, double *__restrict__ scale) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, nc = 6u * n_cp;
    if (i >= nc + 9u) return;
    const double h = i < nc ? ar_band(acc, i, i) : ar_corner(acc, (int) (i - nc), (int) (i - nc));
    scale[i] = jacobi ? 1.0 / (1.0 + sqrt(h)) : 1.0;
}

// 1. the interiors
__global__ __launch_bounds__(64) void arrow_interior_kernel(const double *__restrict__ acc, const double *__restrict__ scale,
                                                            ArrowPlan plan, ArrowLm lm, double *__restrict__ Lfac,
                                                            double *__restrict__ Zfac, int *fail) {
    __shared__ double win[(AR_BW + 6) * (AR_BW + AR_NB)];
    __shared__ double colv[AR_BW + 6];
    __shared__ double pn[(AR_BW - 6) * 6];
    const uint32_t p = blockIdx.x, lane = threadIdx.x;
    const uint32_t lo = 6u * plan.int_lo(p), hi = 6u * plan.int_hi(p), n = hi - lo, nc = 6u * plan.n_cp;
    const bool has_left = p > 0, has_right = p + 1 < plan.P;
    auto load = [&](uint32_t i, uint32_t c) -> double {
        const uint32_t gi = lo + i;
        const double si = scale[gi];
        double v = 0.0;
        if (c < (uint32_t) AR_BW) {                      // band: columns inside the interior
            if (c <= i) {
                v = ar_band(acc, gi, gi - c) * si * scale[gi - c];
                if (c == 0) v += ar_lm_diag(v, lm);
            }
        } else if (c < (uint32_t) (AR_BW + AR_SEP)) {    // left separator: the 18 columns in front of the interior
            const uint32_t a = c - AR_BW, col = lo - AR_SEP + a;
            if (has_left && gi - col < (uint32_t) AR_BW) v = ar_band(acc, gi, col) * si * scale[col];
        } else if (c < (uint32_t) (AR_BW + 2 * AR_SEP)) {   // right separator: its rows reach back into the interior
            const uint32_t a = c - AR_BW - AR_SEP, rw = hi + a;
            if (has_right && rw - gi < (uint32_t) AR_BW) v = ar_band(acc, rw, gi) * si * scale[rw];
        } else if (c < (uint32_t) (AR_BW + 2 * AR_SEP + 9)) {
            const int jx = (int) (c - AR_BW - 2 * AR_SEP);
            v = ar_border(acc, gi, jx) * si * scale[nc + jx];
        } else {
            v = -ar_gc(acc, gi) * si;
        }
        return v;
    };
    const bool ok = band_border_factor<AR_BW, AR_NB, 6, false>(n, load, Lfac + (size_t) lo * AR_BW, Zfac + (size_t) lo * AR_NB, nullptr, win,
                                                                colv, pn);
    if (!ok && lane == 0) atomicExch(fail, 1);
}

// 2. the separators: block tridiagonal in 18-wide blocks, the intrinsics as border
__global__ __launch_bounds__(64) void arrow_reduced_kernel(const double *__restrict__ acc, const double *__restrict__ scale,
                                                           ArrowPlan plan, ArrowLm lm, const double *__restrict__ Gp,
                                                           double *__restrict__ Lred, double *__restrict__ Zred,
                                                           double *__restrict__ Gtot /*[100]*/, int *fail) {
    __shared__ double win[(AR_RBW + AR_SEP) * (AR_RBW + AR_RNB)];
    __shared__ double colv[AR_RBW + AR_SEP];
    __shared__ double pn[(AR_RBW - AR_SEP) * AR_SEP];
    __shared__ double Gred[AR_RNB * AR_RNB];
    const uint32_t lane = threadIdx.x, nc = 6u * plan.n_cp, n = plan.n_red();
    auto G = [&](uint32_t p, uint32_t a, uint32_t b) { return Gp[(size_t) p * AR_NB * AR_NB + a * AR_NB + b]; };
    auto load = [&](uint32_t r, uint32_t c) -> double {
        const uint32_t s = r / AR_SEP, a = r % AR_SEP;           // separator s sits between the partitions s and s + 1
        const uint32_t gi = 6u * plan.int_hi(s) + a;
        const double si = scale[gi];
        double v = 0.0;
        if (c < (uint32_t) AR_RBW) {
            if (c <= r) {
                const uint32_t rc = r - c, s2 = rc / AR_SEP, a2 = rc % AR_SEP;
                if (s2 == s) {   // inside the separator: its own coupling minus both neighbours' Schur contributions
                    const uint32_t gcol = gi - c;
                    v = ar_band(acc, gi, gcol) * si * scale[gcol];
                    if (c == 0) v += ar_lm_diag(v, lm);
                    v -= G(s, AR_SEP + a, AR_SEP + a2) + G(s + 1, a, a2);
                } else if (s2 + 1u == s) {   // the previous separator: coupled through the interior between them only
                    v = -G(s, AR_SEP + a, a2);
                }                            // (band offsets that reach a separator further back: zero)
            }
        } else if (c < (uint32_t) (AR_RBW + 9)) {
            const uint32_t jx = c - AR_RBW;
            v = ar_border(acc, gi, (int) jx) * si * scale[nc + jx] - G(s, AR_SEP + a, 2 * AR_SEP + jx) - G(s + 1, a, 2 * AR_SEP + jx);
        } else {
            v = -ar_gc(acc, gi) * si - G(s, AR_SEP + a, 2 * AR_SEP + 9) - G(s + 1, a, 2 * AR_SEP + 9);
        }
        return v;
    };
    for (uint32_t e = lane; e < (uint32_t) (AR_RNB * AR_RNB); e += 64u) Gred[e] = 0.0;
    ar_wave_sync();
    bool ok = true;
    if (n) ok = band_border_factor<AR_RBW, AR_RNB, AR_SEP, true>(n, load, Lred, Zred, Gred, win, colv, pn);
    ar_wave_sync();
    if (!ok && lane == 0) atomicExch(fail, 1);
    // [Zb z]^T [Zb z] over everything eliminated so far: the interiors' intrinsics / rhs block and the separators'
    for (uint32_t e = lane; e < 100u; e += 64u) {
        const uint32_t i = e / 10u, jx = e % 10u;
        double v = Gred[i * AR_RNB + jx];
        for (uint32_t p = 0; p < plan.P; p++) v += G(p, 2 * AR_SEP + i, 2 * AR_SEP + jx);
        Gtot[e] = v;
    }
}

// 3. the 9 x 9 intrinsics system S = C - Zb^T Zb, b = -g - Zb^T z (Gtot may have been summed over ranks in between), then
// the separators' back substitution L^T y = z - Zb y_intr.  y_red [n_red + 9]: separators, then intrinsics.
__global__ __launch_bounds__(64) void arrow_corner_kernel(const double *__restrict__ acc, const double *__restrict__ scale,
                                                          ArrowPlan plan, ArrowLm lm, const double *__restrict__ Gtot,
                                                          const double *__restrict__ Lred, const double *__restrict__ Zred,
                                                          double *__restrict__ y_red, int *fail, int shared_terms) {
    __shared__ double S[81], bvec[9], yi[9];
    __shared__ int bad;
    const uint32_t lane = threadIdx.x, nc = 6u * plan.n_cp, n = plan.n_red();
    if (lane == 0) bad = 0;
    for (uint32_t e = lane; e < 81u; e += 64u) {
        const int i = (int) (e / 9u), jx = (int) (e % 9u);
        // (shared_terms = 0: distributed ranks other than 0 — C and g of the intrinsics are counted once, by rank 0's Gtot
        //  offset; not used on one GPU)
        double v = (shared_terms ? ar_corner(acc, i, jx) * scale[nc + i] * scale[nc + jx] : 0.0) - Gtot[10 * (i <= jx ? i : jx) + (i <= jx ? jx : i)];
        if (i == jx && shared_terms) v += ar_lm_diag(ar_corner(acc, i, i) * scale[nc + i] * scale[nc + i], lm);
        S[e] = v;
    }
    if (lane < 9u) bvec[lane] = (shared_terms ? -acc[1 + lane] * scale[nc + lane] : 0.0) - Gtot[10 * lane + 9];
    ar_wave_sync();
    if (lane == 0) {   // dense Cholesky 9 x 9 + the two triangular solves: a few hundred flops
        for (int i = 0; i < 9 && !bad; i++)
            for (int jx = 0; jx <= i; jx++) {
                double v = S[9 * i + jx];
                for (int k = 0; k < jx; k++) v -= S[9 * i + k] * S[9 * jx + k];
                if (i == jx) {
                    if (!(v > 0.0)) {
                        bad = 1;
                        break;
                    }
                    S[9 * i + i] = sqrt(v);
                } else {
                    S[9 * i + jx] = v / S[9 * jx + jx];
                }
            }
        if (!bad) {
            for (int i = 0; i < 9; i++) {
                double v = bvec[i];
                for (int k = 0; k < i; k++) v -= S[9 * i + k] * yi[k];
                yi[i] = v / S[9 * i + i];
            }
            for (int i = 8; i >= 0; i--) {
                double v = yi[i];
                for (int k = i + 1; k < 9; k++) v -= S[9 * k + i] * yi[k];
                yi[i] = v / S[9 * i + i];
            }
        } else {
            atomicExch(fail, 1);
        }
    }
    ar_wave_sync();
    if (bad) return;
    if (lane < 9u) y_red[n + lane] = yi[lane];
    // back substitution over the separators: row r needs the AR_RBW - 1 rows after it (y kept in LDS: yr[n]).  The right-hand
    // sides z - Zb y_intr come first, all rows at once; in the chain, lane l holds the term of row r + l + 1 and the factor
    // entries of the NEXT step are requested before the current one is reduced (no memory round trip inside a step).
    extern __shared__ double yr[];
    for (uint32_t r = lane; r < n; r += 64u) {
        double v = Zred[(size_t) r * AR_RNB + 9];
        for (int jx = 0; jx < 9; jx++) v -= Zred[(size_t) r * AR_RNB + jx] * yi[jx];
        yr[r] = v;
    }
    ar_wave_sync();
    {
        const uint32_t k = lane + 1u;
        auto fetch = [&](uint32_t r) -> double {   // lane 63: the diagonal; lanes < AR_RBW - 1: L(r + k, r)
            if (lane == 63u) return Lred[(size_t) r * AR_RBW];
            return (k < (uint32_t) AR_RBW && r + k < n) ? Lred[(size_t) (r + k) * AR_RBW + k] : 0.0;
        };
        double cur_l = n ? fetch(n - 1u) : 0.0;
        for (uint32_t r = n; r-- > 0;) {
            const double nxt_l = r ? fetch(r - 1u) : 0.0;
            double part = (lane != 63u && k < (uint32_t) AR_RBW && r + k < n) ? -cur_l * yr[r + k] : 0.0;
            part = ar_wave_sum_to_last(part);
            if (lane == 63u) yr[r] = (yr[r] + part) / cur_l;
            ar_wave_sync();
            cur_l = nxt_l;
        }
    }
    for (uint32_t r = lane; r < n; r += 64u) y_red[r] = yr[r];
}

// 4. the interiors' back substitution; delta = S y for interiors, separators and intrinsics
__global__ __launch_bounds__(64) void arrow_backsub_kernel(const double *__restrict__ scale, ArrowPlan plan,
                                                           const double *__restrict__ Lfac, const double *__restrict__ Zfac,
                                                           const double *__restrict__ y_red, double *__restrict__ delta) {
    __shared__ double ysl[AR_SEP], ysr[AR_SEP], yin[9];
    extern __shared__ double yl[];   // [interior rows]
    const uint32_t p = blockIdx.x, lane = threadIdx.x, nc = 6u * plan.n_cp, nred = plan.n_red();
    const uint32_t lo = 6u * plan.int_lo(p), hi = 6u * plan.int_hi(p), n = hi - lo;
    const bool has_left = p > 0, has_right = p + 1 < plan.P;
    if (lane < (uint32_t) AR_SEP) {
        ysl[lane] = has_left ? y_red[AR_SEP * (p - 1) + lane] : 0.0;
        ysr[lane] = has_right ? y_red[AR_SEP * p + lane] : 0.0;
    }
    if (lane < 9u) yin[lane] = y_red[nred + lane];
    ar_wave_sync();
    // the parts of the right-hand sides that do not depend on the interior's own unknowns, all rows at once
    for (uint32_t i = lane; i < n; i += 64u) {
        const double *z = Zfac + (size_t) (lo + i) * AR_NB;
        double v = z[2 * AR_SEP + 9];
        for (int a = 0; a < AR_SEP; a++) v -= z[a] * ysl[a] + z[AR_SEP + a] * ysr[a];
        for (int jx = 0; jx < 9; jx++) v -= z[2 * AR_SEP + jx] * yin[jx];
        yl[i] = v;
    }
    ar_wave_sync();
    {
        const uint32_t k = lane + 1u;
        auto fetch = [&](uint32_t i) -> double {   // lane 63: the diagonal; lanes < AR_BW - 1: L(i + k, i)
            if (lane == 63u) return Lfac[(size_t) (lo + i) * AR_BW];
            return (k < (uint32_t) AR_BW && i + k < n) ? Lfac[(size_t) (lo + i + k) * AR_BW + k] : 0.0;
        };
        double cur_l = fetch(n - 1u);
        for (uint32_t i = n; i-- > 0;) {
            const double nxt_l = i ? fetch(i - 1u) : 0.0;
            double part = (lane != 63u && k < (uint32_t) AR_BW && i + k < n) ? -cur_l * yl[i + k] : 0.0;
            part = ar_wave_sum_to_last(part);
            if (lane == 63u) yl[i] = (yl[i] + part) / cur_l;
            ar_wave_sync();
            cur_l = nxt_l;
        }
    }
    for (uint32_t i = lane; i < n; i += 64u) delta[lo + i] = yl[i] * scale[lo + i];
    if (has_right && lane < (uint32_t) AR_SEP) delta[hi + lane] = ysr[lane] * scale[hi + lane];   // this partition's right separator
    if (p == 0 && lane < 9u) delta[nc + lane] = yin[lane] * scale[nc + lane];
}

// x (+) delta: intrinsics and translations add, the rotation control points take the local parameterisation's Plus
template <bool SO3>
__global__ void lm_plus_kernel(const double *__restrict__ x, const double *__restrict__ delta, uint32_t n_cp, double *__restrict__ out) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nc = 6 * (size_t) n_cp;
    if (c < n_cp) {
        double q[4], d[3], o[4];
        for (int k = 0; k < 4; k++) q[k] = x[9 + 4 * (size_t) c + k];
        for (int k = 0; k < 3; k++) d[k] = delta[6 * (size_t) c + k];
        if (SO3) so3_plus(q, d, o); else quaternion_plus(q, d, o);
        for (int k = 0; k < 4; k++) out[9 + 4 * (size_t) c + k] = o[k];
        for (int k = 0; k < 3; k++)
            out[9 + 4 * (size_t) n_cp + 3 * (size_t) c + k] = x[9 + 4 * (size_t) n_cp + 3 * (size_t) c + k] + delta[6 * (size_t) c + 3 + k];
    } else if (c < n_cp + 9u) {
        const uint32_t i = c - n_cp;
        out[i] = x[i] + delta[nc + i];
    }
}

// g^T d, d^T A d (unscaled system), |d|^2 and |x|^2: what the step acceptance needs, four sums -> sums[0 .. 3]
__global__ void arrow_quad_kernel(const double *__restrict__ acc, const double *__restrict__ delta, const double *__restrict__ x,
                                  uint32_t n_cp, uint32_t n_params, double *__restrict__ sums) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, nc = 6u * n_cp;
    double g = 0, h = 0, s2 = 0, x2 = 0;
    if (i < nc) {
        const double di = delta[i];
        g = ar_gc(acc, i) * di;
        double row = ar_band(acc, i, i) * di;
        const uint32_t kmax = i < (uint32_t) (AR_BW - 1) ? i : (uint32_t) (AR_BW - 1);
        for (uint32_t k = 1; k <= kmax; k++) row += 2.0 * ar_band(acc, i, i - k) * delta[i - k];
        h = di * row;
        for (int jx = 0; jx < 9; jx++) h += 2.0 * di * ar_border(acc, i, jx) * delta[nc + jx];
        s2 = di * di;
    } else if (i < nc + 9u) {
        const int a = (int) (i - nc);
        const double di = delta[i];
        g = acc[1 + a] * di;
        for (int jx = 0; jx < 9; jx++) h += di * ar_corner(acc, a, jx) * delta[nc + jx];
        s2 = di * di;
    }
    if (i < n_params) x2 = x[i] * x[i];
    for (int o = 32; o > 0; o >>= 1) {
        g += __shfl_xor(g, o, 64);
        h += __shfl_xor(h, o, 64);
        s2 += __shfl_xor(s2, o, 64);
        x2 += __shfl_xor(x2, o, 64);
    }
    if ((threadIdx.x & 63u) == 0) {
        atomicAdd(&sums[0], g);
        atomicAdd(&sums[1], h);
        atomicAdd(&sums[2], s2);
        atomicAdd(&sums[3], x2);
    }
}

// max |gradient| of a normal-equation buffer -> *out (as the bits of a non-negative double: atomicMax on the integer)
__global__ void arrow_gmax_kernel(const double *__restrict__ acc, uint32_t n_cp, unsigned long long *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, nc = 6u * n_cp;
    double v = 0.0;
    if (i < nc) v = fabs(ar_gc(acc, i));
    else if (i < nc + 9u) v = fabs(acc[1 + (i - nc)]);
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    if ((threadIdx.x & 63u) == 0) atomicMax(out, (unsigned long long) __double_as_longlong(v));
}

}  // namespace ecal
