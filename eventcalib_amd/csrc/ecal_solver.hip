// Continuous-time calibration solve: per-event residual / Jacobian and the stacked normal equations
// on the GPU, Levenberg-Marquardt and the block-banded arrow Cholesky on the host.
//
// Replaces, for the quaternion-spline variant (useSO3 = 0, the shipped default):
//   EventCalibSpline::optimize's Ceres problem (event_camera_calib/src/EventCalibSpline.cpp:196-247):
//     one residual block [9,4,4,4,4,3,3,3,3] per associated event, HuberLoss(0.2 R),
//     EigenQuaternionParameterization, SPARSE_NORMAL_CHOLESKY, tolerances 1e-10, 50 iterations;
//   CalibReprojectionError::operator() (EventCalibSpline.hpp:158-229) — see spline_residual.hpp.
// Residuals are sorted by (segment, time); a "chunk" is a run of residuals inside one knot span, so all
// its rows share the same 33 columns and J^T J of the chunk is one dense 34x34 (33 + residual) Gram
// matrix: a genuine dense contraction, accumulated on the matrix cores with v_mfma_f64_4x4x4_4b_f64 (nine 4-column tiles of
// the padded 36 columns, 45 tile pairs, 16 rows per round — see normal_eq_kernel).  History of the alternatives, each with its
// measurement: v_mfma_f64_16x16x4_f64 (34 -> 48 columns, 2x the multiply-adds: 4.63 ms per evaluation against 3.59,
// profiles/experiments/r02_normal_eq_mfma_f64.patch); register-tiled v_fma_f64 (6x6 tiles, rounds 2 - 5: 2.28 ms) and
// producer / consumer waves on those tiles (2.70 ms) against this kernel's 1.71 ms at 45 M residuals
// (profiles/experiments/r06_normal_eq_fma_tiles_and_two_roles.patch, profiles/r06_ne_mfma_vs_fma.txt).
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <limits>
#include <sched.h>
#include "ecal_ctx.hpp"
#include "spline_residual.hpp"
#include "arrow_layout.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <vector>

namespace ecal {

struct ResRecord {  // 32 bytes per residual: the algorithmic traffic unit of SURVEY §8(d)
    double u, v, t;
    uint32_t lm, seg;
};
struct Chunk {
    uint32_t start, count, seg, span;
};

constexpr int NE_T = 256;   // threads per workgroup = rows per batch
constexpr int NE_LD = 36;   // padded row: 33 Jacobian entries, the residual, 2 zeros
// Gram accumulation: 4-column tiles of the padded row (v_mfma_f64_4x4x4_4b_f64 multiplies four 4x4 blocks per instruction)
constexpr int NE_TW = 4, NE_TG = NE_LD / NE_TW, NE_TILES = NE_TG * (NE_TG + 1) / 2, NE_WAVES = NE_T / 64;
constexpr uint32_t NE_CHUNK = 16384;  // residuals per workgroup (one span): few, long chunks keep the FP64 atomics rare
constexpr uint32_t NE_REPL = 64;      // replicas of the shared head (cost, intrinsics block) that the chunks add into

// (accumulation buffer layout: ACC_HEAD, ACC_PER_CP — arrow_layout.hpp)

__device__ __forceinline__ void local_to_unknown(int li, uint32_t c0, bool &is_intr, uint32_t &cp, uint32_t &comp) {
    if (li < 9) {
        is_intr = true;
        cp = 0;
        comp = (uint32_t) li;
    } else if (li < 21) {
        is_intr = false;
        cp = c0 + (uint32_t) (li - 9) / 3u;
        comp = (uint32_t) (li - 9) % 3u;
    } else {
        is_intr = false;
        cp = c0 + (uint32_t) (li - 21) / 3u;
        comp = 3u + (uint32_t) (li - 21) % 3u;
    }
}

// (profiling builds: -DECAL_NE_SKIP_P1 / -DECAL_NE_SKIP_P2 drop the residual code / the Gram accumulation: the phase split of
// profiles/r06_notes.md.  Results are meaningless in those builds.)
// Row index swizzle of the wave's column-major slab (see normal_eq_kernel)
__host__ __device__ constexpr int ne_swz(int c) { return 16 * (c & 1) ^ 2 * ((c >> 1) & 1); }

// A streamed evaluation (ecal_solver_solve, one rank, a long spline): the host factorises the interiors of its partition of the
// control points (arrow_host_parts.hpp) WHILE the kernel is still accumulating the later ones.  The chunks are ordered by knot
// span, i.e. by control point; group g = the chunks whose span's last control point lies in [cut[g], cut[g + 1]).  A chunk
// with last control point s adds to the records s - 3 .. s, so the records of interior g ([cut[g], cut[g + 1] - 3)) are touched
// by group g alone and the three records of the separator behind it by groups g and g + 1.  The workgroup that finishes a group
// (a counter per group) copies the interior's records into the host's pinned buffer and raises the group's flag there; the
// second of the two groups beside a separator to finish does the same for the separator's records.  The counters are restored
// by the workgroup that zeroes them: nothing to prepare per launch.
// (NE_MAX_GROUPS: arrow_layout.hpp)
struct NeProgress {
    uint32_t n_groups, n_cp;
    uint32_t cut[NE_MAX_GROUPS + 1];      // cut[n_groups] = n_cp
    uint32_t init[2 * NE_MAX_GROUPS];     // [g]: chunks of group g; [NE_MAX_GROUPS + b]: groups with chunks beside separator b
    uint32_t *left;                       // the running counters, same layout (device memory)
    double *host_acc;                     // the pinned accumulation buffer as the device sees it
    uint32_t *host_flag;                  // pinned; [g] / [NE_MAX_GROUPS + b] = number of the evaluation that delivered them
};

// records [r_lo, r_hi) of the accumulation buffer to the host's copy; the values were added by other workgroups' atomics on
// any of the eight XCDs: agent-scope loads (a plain load could be served from this XCD's L2)
__device__ __forceinline__ void ne_copy_records(const double *accum, double *host, uint32_t r_lo, uint32_t r_hi, int tid, int nthreads) {
    const size_t lo = ACC_HEAD + ACC_PER_CP * (size_t) r_lo, hi = ACC_HEAD + ACC_PER_CP * (size_t) r_hi;
    for (size_t i = lo + (size_t) tid; i < hi; i += 8 * (size_t) nthreads) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const size_t k = i + (size_t) u * nthreads;
            v[u] = k < hi ? __hip_atomic_load(accum + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const size_t k = i + (size_t) u * nthreads;
            if (k < hi) host[k] = v[u];
        }
    }
}

// the end of a chunk's workgroup in a streamed evaluation (NeProgress): the group's (and its separators') records to the host
// when this chunk is the last one of the group.  Called by every thread of the workgroup (nthreads of them) after its adds.
__device__ __forceinline__ void ne_publish_progress(const NeProgress *__restrict__ prog, uint32_t epoch, const double *accum, uint32_t c0, int tid,
                                                    int nthreads) {
    __shared__ uint32_t fin[3];
    __threadfence();
    __syncthreads();   // every thread's adds are out
    const uint32_t P = prog->n_groups;
    if (tid == 0) {
        const uint32_t sg = c0 + 3;
        uint32_t g = 0;
        while (g + 1 < P && prog->cut[g + 1] <= sg) g++;
        uint32_t *left = prog->left;
        const bool last = atomicSub(&left[g], 1u) == 1u;
        fin[0] = last ? g + 1 : 0;
        fin[1] = fin[2] = 0;
        if (last) {
            left[g] = prog->init[g];
            if (g > 0 && atomicSub(&left[NE_MAX_GROUPS + g - 1], 1u) == 1u) {
                fin[1] = 1;
                left[NE_MAX_GROUPS + g - 1] = prog->init[NE_MAX_GROUPS + g - 1];
            }
            if (g + 1 < P && atomicSub(&left[NE_MAX_GROUPS + g], 1u) == 1u) {
                fin[2] = 1;
                left[NE_MAX_GROUPS + g] = prog->init[NE_MAX_GROUPS + g];
            }
        }
    }
    __syncthreads();
    if (fin[0]) {
        __threadfence();
        const uint32_t g = fin[0] - 1;
        const uint32_t c_lo = prog->cut[g], c_next = prog->cut[g + 1];
        double *host = prog->host_acc;
        ne_copy_records(accum, host, c_lo, g + 1 < P ? c_next - 3 : c_next, tid, nthreads);
        if (fin[1]) ne_copy_records(accum, host, c_lo - 3, c_lo, tid, nthreads);
        if (fin[2]) ne_copy_records(accum, host, c_next - 3, c_next, tid, nthreads);
        __threadfence_system();
        __syncthreads();
        if (tid == 0) {
            uint32_t *flag = prog->host_flag;
            __hip_atomic_store(&flag[g], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            if (fin[1]) __hip_atomic_store(&flag[NE_MAX_GROUPS + g - 1], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            if (fin[2]) __hip_atomic_store(&flag[NE_MAX_GROUPS + g], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// WITH_JAC = false: cost only (no rows, no LDS, no tiles) — its own, small instantiation (69 VGPRs against 233)
template <bool SO3, bool WITH_JAC, bool FISHEYE = false>
__global__ __launch_bounds__(NE_T, SO3 ? 1 : 2) void normal_eq_kernel(const ResRecord *__restrict__ rec,
                                                         const Chunk *__restrict__ chunks,
                                                         const double *__restrict__ knots,
                                                         const uint32_t *__restrict__ knot_off,
                                                         const uint32_t *__restrict__ cp_off,
                                                         const double *__restrict__ params, uint32_t n_cp_total,
                                                         const double *__restrict__ landmarks, double radius,
                                                         double huber_a, double *__restrict__ accum,
                                                         double *__restrict__ heads, const NeProgress *__restrict__ prog,
                                                         uint32_t epoch) {
    constexpr bool with_jac = WITH_JAC;
    extern __shared__ __attribute__((aligned(16))) double rows[];  // with_jac: [NE_WAVES][NE_LD columns][64 rows]
    __shared__ double red[NE_T / 64];
    const Chunk ch = chunks[blockIdx.x];
    const int tid = threadIdx.x;
    // every chunk touches the cost and the 9x9 intrinsics block: spread those adds over NE_REPL copies
    double *const head = heads + (size_t) (blockIdx.x % NE_REPL) * ACC_HEAD;
    const double *kn = knots + knot_off[ch.seg];
    const uint32_t c0 = cp_off[ch.seg] + ch.span - 3;
    const double *intr = params;
    const double *qall = params + 9;
    const double *tall = params + 9 + 4 * (size_t) n_cp_total;
    double q[4][4], t[4][3], pin[9], binv[6];
    for (int i = 0; i < 9; i++) pin[i] = intr[i];
    spline_span_inverses(kn, ch.span, binv);           // uniform over the chunk: 6 + 2 divisions per thread, not per residual
    const double ifx = 1.0 / pin[0], ify = 1.0 / pin[1], inv_huber_a = 1.0 / huber_a;
    for (int j = 0; j < 4; j++) {
        for (int k = 0; k < 4; k++) q[j][k] = qall[4 * (size_t) (c0 + j) + k];
        for (int k = 0; k < 3; k++) t[j][k] = tall[3 * (size_t) (c0 + j) + k];
    }
    // Gram accumulation on the matrix cores.  v_mfma_f64_4x4x4_4b_f64 multiplies four independent 4x4x4 blocks: block blk takes
    // rows 4 blk .. 4 blk + 3 of a 16-row step, and all four blocks multiply the same pair of column tiles (ta, tb), so lane
    // (x, blk, k) = (l & 3, (l >> 2) & 3, l >> 4) supplies J[row 4 blk + k][column 4 ta + x] as A[blk][i = x][k] and
    // J[row 4 blk + k][4 tb + x] as B[blk][k][j = x]: ONE register per column tile serves as the left and as the right operand
    // (lane maps: profiles/r05_mfma_f64_probe.txt).  Nine 8-byte LDS reads per lane and 45 instructions per 16 rows; 45
    // accumulators per lane (D[blk][i][j] of pair p in lane j + 4 blk + 16 i), summed over blk at the end of the chunk.
    // The rows go through THIS WAVE's slab of the row buffer, column-major ([column][row of the wave's 64]) with the row index
    // XOR-ed by ne_swz(column): lane l writes its row's entry of column c at c * 64 + (l ^ swz(c)) — a permutation of 64
    // consecutive doubles, conflict-free — and the operand read above puts the 32 lanes of either half-wave on 32 different
    // 8-byte bank pairs.  No other wave touches the slab: the batch loop has no workgroup barrier.
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    double acc[NE_TILES];
#pragma unroll
    for (int i = 0; i < NE_TILES; i++) acc[i] = 0.0;
    double cost = 0.0;
    double *const slab = rows + (size_t) wave * (NE_LD * 64);
    if constexpr (with_jac) {
        slab[34 * 64 + lane] = 0.0;     // the two padding columns of tile 8: written once
        slab[35 * 64 + lane] = 0.0;
    }

    // the record of the NEXT batch is asked for before this batch's residual is evaluated: its trip to HBM (the only one a
    // batch makes) runs behind ~2000 cycles of arithmetic instead of in front of them
    ResRecord e_next = rec[ch.start + min((uint32_t) tid, ch.count - 1u)];
    for (uint32_t b0 = 0; b0 < ch.count; b0 += NE_T) {
        const uint32_t k = b0 + tid;
        double J[RES_NJ];
        double r = 0.0, sc = 0.0;
        const ResRecord e = e_next;
        e_next = rec[ch.start + min(k + (uint32_t) NE_T, ch.count - 1u)];
        // (no branch around the residual code: a lane past the chunk's end — the last batch only — evaluates the chunk's last
        // record again, which it holds anyway (clamped load), and its row is scaled by zero: the row's 34 stores need no selects)
        const bool live = k < ch.count;
#ifdef ECAL_NE_SKIP_P1
        {
            r = e.u;
            sc = live ? e.v : 0.0;
            for (int i = 0; i < RES_NJ; i++) J[i] = e.t + i;
            cost += live ? r : 0.0;
        }
        if (false) {
#else
        {
#endif
            ResidualInput in;
            in.u = e.u;
            in.v = e.v;
            in.lmx = landmarks[3 * (size_t) e.lm];
            in.lmy = landmarks[3 * (size_t) e.lm + 1];
            in.lmz = landmarks[3 * (size_t) e.lm + 2];
            in.radius = radius;
            in.ifx = ifx;
            in.ify = ify;
            spline_basis_inv(kn, ch.span, binv, e.t, in.b);
            double hr = 0.0;
            if constexpr (with_jac) {   // the row comes out scaled by sqrt(rho') (ResidualInput)
                in.huber_a = huber_a;
                in.dead = !live;
                in.inv_huber_a = inv_huber_a;
                in.sc_out = &sc;
                in.half_rho_out = &hr;
            }
            r = SO3 ? spline_residual_so3<FISHEYE>(in, pin, q, t, with_jac ? J : nullptr)
                    : spline_residual<FISHEYE>(in, pin, q, t, with_jac ? J : nullptr);
            if constexpr (!with_jac) sc = huber_scale(r, huber_a, &hr, inv_huber_a);
            cost += live ? hr : 0.0;
        }
        if constexpr (with_jac) {
#pragma unroll
            for (int i = 0; i < RES_NJ; i++) slab[i * 64 + (lane ^ ne_swz(i))] = J[i];
            slab[33 * 64 + (lane ^ ne_swz(33))] = r * sc;
            // the slab is this wave's alone: no workgroup barrier.  A wave's LDS operations complete in order; the fence keeps the
            // compiler from moving the reads above the writes (other lanes' data) or the next batch's writes above these reads.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#ifndef ECAL_NE_SKIP_P2
            const uint32_t nrow = min((uint32_t) NE_T, ch.count - b0);
            const int x = lane & 3, r0 = ((lane >> 2) & 3) * 4 + (lane >> 4);
            const int pb = x * 64 + (r0 ^ (x & 2));                   // (row ^ swz) = 16 (s ^ (x & 1)) + (r0 ^ (x & 2))
            const double *const base_e = slab + pb + 16 * (x & 1);      // even steps (s = 0, 2): + 32 (s >> 1)
            const double *const base_o = slab + pb + 16 * (1 - (x & 1));  // odd steps
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) {
                if ((uint32_t) (64 * wave + 16 * s4) >= nrow) break;     // wave-uniform: only the chunk's last batch
                const double *const bs = ((s4 & 1) ? base_o : base_e) + 32 * (s4 >> 1);
                double R[NE_TG];
#pragma unroll
                for (int t9 = 0; t9 < NE_TG; t9++) R[t9] = bs[t9 * 256];
                int p = 0;
#pragma unroll
                for (int ta = 0; ta < NE_TG; ta++)
#pragma unroll
                    for (int tb = ta; tb < NE_TG; tb++, p++) acc[p] = __builtin_amdgcn_mfma_f64_4x4x4f64(R[ta], R[tb], acc[p], 0, 0, 0);
            }
#endif
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    // cost: block reduction, one atomic per workgroup
    for (int o = 32; o > 0; o >>= 1) cost += __shfl_down(cost, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = cost;
    __syncthreads();
    if (tid == 0) {
        double c = 0;
        for (int w = 0; w < NE_T / 64; w++) c += red[w];
        atomicAdd(&head[0], c);
    }
    if (!with_jac) return;
    // the waves' partial tiles are summed in LDS (the row buffer is free now) so that ONE set of FP64 atomics per chunk goes out
    // (~600 instead of 4 x ~600)
    {
        constexpr int TSZ = NE_TW * NE_TW, NPART = NE_TILES * TSZ;
        double *part = rows;                         // [NE_WAVES][NE_TILES * TSZ]
        double *total = rows;                        // (in place of wave 0's slice: entry e is read and written by ONE thread)
        static_assert(NE_WAVES * NPART <= NE_T * NE_LD, "the row buffer holds the partial tiles");
        __syncthreads();
        // D[blk][i][j] of tile pair p sits in lane j + 4 blk + 16 i: the four blocks (this wave's rows 4 blk + k of every step)
        // are summed across lane bits 2-3, block 0's lanes store entry (i, j) of the wave's partial tile
#pragma unroll
        for (int p = 0; p < NE_TILES; p++) {
            double v = acc[p];
            v += __shfl_xor(v, 4, 64);
            v += __shfl_xor(v, 8, 64);
            if ((lane & 12) == 0) part[(size_t) wave * NPART + p * TSZ + 4 * (lane >> 4) + (lane & 3)] = v;
        }
        __syncthreads();
        for (int e = tid; e < NPART; e += NE_T) {    // every thread sums a few entries over the groups
            double v = 0.0;
#pragma unroll
            for (int g = 0; g < NE_WAVES; g++) v += part[(size_t) g * NPART + e];
            total[e] = v;
        }
        __syncthreads();
        // flush: one entry of the summed tiles per thread and turn (not a tile per thread: nothing of acc[] stays live)
        for (int e = tid; e < NPART; e += NE_T) {
            const double v = total[e];
            if (v == 0.0) continue;
            const int tile = e / TSZ, x = (e % TSZ) / NE_TW, y = e % NE_TW;
            int fi = 0, rem = tile;   // tile -> (fi, fj), fi <= fj, rows of the upper triangle
            for (fi = 0; fi < NE_TG; fi++) {
                if (rem < NE_TG - fi) break;
                rem -= NE_TG - fi;
            }
            const int fj = fi + rem;
            const int li = NE_TW * fi + x, lj = NE_TW * fj + y;
            if (li > lj || lj >= 34 || li >= 33) continue;
            bool ia, ib;
            uint32_t ca, ka, cb, kb;
            local_to_unknown(li, c0, ia, ca, ka);
            if (lj == 33) {  // gradient J^T r
                if (ia) atomicAdd(&head[1 + ka], v);
                else atomicAdd(&accum[ACC_HEAD + ACC_PER_CP * (size_t) ca + ka], v);
                continue;
            }
            local_to_unknown(lj, c0, ib, cb, kb);
            if (ia && ib) {
                atomicAdd(&head[10 + 9 * ka + kb], v);
            } else if (ia) {  // intrinsics x control point
                atomicAdd(&accum[ACC_HEAD + ACC_PER_CP * (size_t) cb + 6 + 9 * kb + ka], v);
            } else {
                if (ca > cb || (ca == cb && ka > kb)) {  // keep blocks (c, c+d), d >= 0; diagonal upper
                    const uint32_t tc = ca, tk = ka;
                    ca = cb;
                    ka = kb;
                    cb = tc;
                    kb = tk;
                }
                atomicAdd(&accum[ACC_HEAD + ACC_PER_CP * (size_t) ca + 60 + 36 * (cb - ca) + 6 * ka + kb], v);
            }
        }
    }
    if (prog) ne_publish_progress(prog, epoch, accum, c0, tid, NE_T);
}

template <bool SO3, bool FISHEYE = false>
__global__ __launch_bounds__(NE_T) void residual_rows_kernel(const ResRecord *__restrict__ rec, const Chunk *__restrict__ chunks,
                                                            const double *__restrict__ knots, const uint32_t *__restrict__ knot_off,
                                                            const uint32_t *__restrict__ cp_off, const double *__restrict__ params,
                                                            uint32_t n_cp_total, const double *__restrict__ landmarks, double radius,
                                                            double *__restrict__ r_out, double *__restrict__ J_out,
                                                            uint32_t *__restrict__ cp0_out) {
    const Chunk ch = chunks[blockIdx.x];
    const double *kn = knots + knot_off[ch.seg];
    const uint32_t c0 = cp_off[ch.seg] + ch.span - 3;
    const double *qall = params + 9;
    const double *tall = params + 9 + 4 * (size_t) n_cp_total;
    double q[4][4], t[4][3], pin[9], binv[6];
    for (int i = 0; i < 9; i++) pin[i] = params[i];
    spline_span_inverses(kn, ch.span, binv);
    const double ifx = 1.0 / pin[0], ify = 1.0 / pin[1];
    for (int j = 0; j < 4; j++) {
        for (int k = 0; k < 4; k++) q[j][k] = qall[4 * (size_t) (c0 + j) + k];
        for (int k = 0; k < 3; k++) t[j][k] = tall[3 * (size_t) (c0 + j) + k];
    }
    for (uint32_t k = threadIdx.x; k < ch.count; k += NE_T) {
        const size_t at = (size_t) ch.start + k;
        const ResRecord e = rec[at];
        ResidualInput in;
        in.u = e.u;
        in.v = e.v;
        in.lmx = landmarks[3 * (size_t) e.lm];
        in.lmy = landmarks[3 * (size_t) e.lm + 1];
        in.lmz = landmarks[3 * (size_t) e.lm + 2];
        in.radius = radius;
        in.ifx = ifx;
        in.ify = ify;
        spline_basis_inv(kn, ch.span, binv, e.t, in.b);
        double J[RES_NJ];
        const double r = SO3 ? spline_residual_so3<FISHEYE>(in, pin, q, t, J_out ? J : nullptr)
                             : spline_residual<FISHEYE>(in, pin, q, t, J_out ? J : nullptr);
        r_out[at] = r;
        if (J_out)
            for (int i = 0; i < RES_NJ; i++) J_out[at * RES_NJ + i] = J[i];
        if (cp0_out) cp0_out[at] = c0;
    }
}

// accum[0..91) = sum over the replicas
__global__ void reduce_heads_kernel(const double *__restrict__ heads, double *__restrict__ accum, uint32_t n_out) {
    const uint32_t i = threadIdx.x;
    if (i >= n_out) return;
    double v = 0.0;
    for (uint32_t r = 0; r < NE_REPL; r++) v += heads[(size_t) r * ACC_HEAD + i];
    accum[i] = v;
}

}  // namespace ecal

using namespace ecal;

struct ecal_solver {
    ecal_ctx *ctx = nullptr;
    uint64_t n_res = 0;
    uint32_t n_cp = 0, n_seg = 0, n_chunks = 0;
    double radius = 0, huber_a = 0;
    bool use_so3 = false;  // cumulative SO3 spline + LocalParameterizationSO3 instead of the quaternion spline
    bool fisheye = false;  // camera_model == ECAL_CAMERA_FISHEYE
    std::vector<uint32_t> cp_off, knot_off;
    std::vector<double> knots;
    ResRecord *d_rec = nullptr;
    Chunk *d_chunks = nullptr;
    double *d_knots = nullptr, *d_landmarks = nullptr, *d_params = nullptr, *d_accum = nullptr, *d_heads = nullptr;
    uint32_t *d_knot_off = nullptr, *d_cp_off = nullptr;
    // ecal_solver_solve's pinned staging (kept: pinning 3 MB per solve costs more than an LM iteration) and the streamed
    // evaluation's progress block (NeProgress)
    double *h_acc = nullptr, *h_x = nullptr;
    uint32_t *h_flag = nullptr, *d_left = nullptr;
    NeProgress *d_prog = nullptr;
    NeProgress prog{};             // the host's copy (prog.n_groups = 0: not set up)
    std::shared_ptr<void> host_pool;   // ecal_solver_solve's worker threads (HostPool), parked between solves
    int host_pool_workers = -1;
    uint32_t stream_epoch = 0;
    uint32_t last_solve[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // ecal_debug_solver_last_solve: how the last ecal_solver_solve ran
    size_t n_params() const { return 9 + 7 * (size_t) n_cp; }
    size_t n_accum() const { return ACC_HEAD + ACC_PER_CP * (size_t) n_cp; }
};

extern "C" void ecal_solver_destroy(ecal_solver *s) {
    if (!s) return;
    (void) hipSetDevice(s->ctx->device);
    void *ptrs[] = {s->d_rec, s->d_chunks, s->d_knots, s->d_landmarks, s->d_params, s->d_accum, s->d_heads, s->d_knot_off, s->d_cp_off,
                    s->d_left, s->d_prog};
    for (void *p : ptrs)
        if (p) (void) hipFree(p);
    void *pinned[] = {s->h_acc, s->h_x, s->h_flag};
    for (void *p : pinned)
        if (p) (void) hipHostFree(p);
    delete s;
}

extern "C" size_t ecal_solver_normal_size(const ecal_solver *s) { return s ? s->n_accum() : 0; }
extern "C" size_t ecal_solver_param_size(const ecal_solver *s) { return s ? s->n_params() : 0; }
extern "C" uint32_t ecal_solver_num_chunks(const ecal_solver *s) { return s ? s->n_chunks : 0; }

extern "C" int ecal_solver_create(ecal_ctx *ctx, const ecal_spline_problem *p, ecal_solver **out) {
    if (!ctx || !p || !out) return ECAL_ERR_INVALID;
    *out = nullptr;
    if (p->n_segments < 1 || !p->seg_cp_off || !p->knots || !p->landmarks || (p->n_res && (!p->obs || !p->time || !p->lm_id))) {
        ctx->last_error = "null pointer / no segment";
        return ECAL_ERR_INVALID;
    }
    if (p->n_res > 0xFFFFFFFFull) return ECAL_ERR_RANGE;
    ecal_solver *s = new (std::nothrow) ecal_solver;
    if (!s) return ECAL_ERR_NOMEM;
    s->ctx = ctx;
    s->n_res = p->n_res;
    s->n_seg = p->n_segments;
    s->radius = p->circle_radius;
    s->huber_a = p->huber_a;
    s->use_so3 = p->use_so3 != 0;
    s->fisheye = p->camera_model == ECAL_CAMERA_FISHEYE;
    s->cp_off.assign(p->seg_cp_off, p->seg_cp_off + p->n_segments + 1);
    s->n_cp = s->cp_off[p->n_segments];
    s->knot_off.resize(p->n_segments + 1);
    for (uint32_t g = 0; g <= p->n_segments; g++) s->knot_off[g] = s->cp_off[g] + 4 * g;
    s->knots.assign(p->knots, p->knots + s->knot_off[p->n_segments]);
    for (uint32_t g = 0; g < p->n_segments; g++) {
        if (s->cp_off[g + 1] - s->cp_off[g] < 4) {
            ctx->last_error = "a spline segment needs at least degree + 1 = 4 control points";
            delete s;
            return ECAL_ERR_INVALID;
        }
    }
    // chunk table: residuals are sorted by (segment, time); cut at knot-span boundaries
    std::vector<ResRecord> recs(p->n_res);
    std::vector<Chunk> chunks;
    uint64_t i = 0;
    for (uint32_t g = 0; g < p->n_segments; g++) {
        const double *kn = s->knots.data() + s->knot_off[g];
        const uint32_t ncp = s->cp_off[g + 1] - s->cp_off[g];
        uint64_t e = i;
        while (e < p->n_res && (p->seg_id ? p->seg_id[e] : 0u) == g) e++;
        for (uint64_t k = i; k < e; k++) {
            if ((k > i && p->time[k] < p->time[k - 1]) || p->time[k] < kn[3] || p->time[k] > kn[ncp]) {
                ctx->last_error = "residual times must be sorted inside a segment and lie inside its knot range";
                delete s;
                return ECAL_ERR_INVALID;
            }
            if (p->lm_id[k] >= p->n_landmarks) {
                ctx->last_error = "landmark id out of range";
                delete s;
                return ECAL_ERR_INVALID;
            }
        }
        uint64_t a = i;
        for (uint32_t span = 3; span < ncp && a < e; span++) {
            uint64_t b;
            if (span == ncp - 1) {
                b = e;
            } else {
                b = std::lower_bound(p->time + a, p->time + e, kn[span + 1]) - p->time;  // [u_span, u_span+1)
            }
            // a span's residuals in EQUAL chunks of at most NE_CHUNK (whole batches of NE_T rows): with fixed-size chunks and a
            // remainder the workgroups alternate long / short (64 and 25 batches on the benchmark problem) and the launch ran a
            // third longer than with either 32 + 32 + 25 or one chunk of 88 batches (3.52 against 2.66 / 2.65 ms)
            if (b > a) {
                const uint64_t m = b - a, parts = (m + NE_CHUNK - 1) / NE_CHUNK;
                const uint64_t per = ((m + parts - 1) / parts + NE_T - 1) / NE_T * NE_T;
                for (uint64_t c = a; c < b; c += per) chunks.push_back(Chunk{(uint32_t) c, (uint32_t) std::min<uint64_t>(per, b - c), g, span});
            }
            a = b;
        }
        i = e;
    }
    if (i != p->n_res) {
        ctx->last_error = "seg_id must be non-decreasing and < n_segments";
        delete s;
        return ECAL_ERR_INVALID;
    }
    for (uint64_t k = 0; k < p->n_res; k++) {
        recs[k].u = p->obs[2 * k];
        recs[k].v = p->obs[2 * k + 1];
        recs[k].t = p->time[k];
        recs[k].lm = p->lm_id[k];
        recs[k].seg = p->seg_id ? p->seg_id[k] : 0u;
    }
    s->n_chunks = (uint32_t) chunks.size();
    hipError_t e = hipSetDevice(ctx->device);
    auto up = [&](void **dst, const void *src, size_t bytes) {
        if (e != hipSuccess) return;
        e = hipMalloc(dst, bytes ? bytes : 16);
        if (e == hipSuccess && bytes) e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
    };
    up((void **) &s->d_rec, recs.data(), recs.size() * sizeof(ResRecord));
    up((void **) &s->d_chunks, chunks.data(), chunks.size() * sizeof(Chunk));
    up((void **) &s->d_knots, s->knots.data(), s->knots.size() * sizeof(double));
    up((void **) &s->d_landmarks, p->landmarks, 3 * (size_t) p->n_landmarks * sizeof(double));
    up((void **) &s->d_knot_off, s->knot_off.data(), s->knot_off.size() * sizeof(uint32_t));
    up((void **) &s->d_cp_off, s->cp_off.data(), s->cp_off.size() * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **) &s->d_params, s->n_params() * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **) &s->d_accum, s->n_accum() * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **) &s->d_heads, NE_REPL * ACC_HEAD * sizeof(double));
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&normal_eq_kernel<false, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int) (NE_T * NE_LD * sizeof(double)));
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&normal_eq_kernel<true, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int) (NE_T * NE_LD * sizeof(double)));
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&normal_eq_kernel<false, true, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int) (NE_T * NE_LD * sizeof(double)));
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&normal_eq_kernel<true, true, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int) (NE_T * NE_LD * sizeof(double)));
    if (e != hipSuccess) {
        ctx->last_error = std::string("ecal_solver_create: ") + hipGetErrorString(e);
        ecal_solver_destroy(s);
        return e == hipErrorOutOfMemory ? ECAL_ERR_NOMEM : ECAL_ERR_HIP;
    }
    *out = s;
    return ECAL_OK;
}

// ---- the problem built in place from DEVICE arrays (the association's outputs) ---------------------------------------------
// EventCalibSpline::optimize adds a residual block per associated event where it finds it (EventCalibSpline.cpp:181-235);
// here the association (ecal_associate_ranges_dev) leaves obs / time / lm_id / seg_id in HBM and the solver's records and
// chunk table are made from them by three small kernels — the 28 bytes per residual never cross PCIe.
namespace {

// flags: 1 residual out of time order / outside its segment's knot range, 2 landmark id out of range, 4 segment ids not
// non-decreasing or >= n_segments
__global__ void solver_pack_kernel(const double *__restrict__ obs, const double *__restrict__ time, const uint32_t *__restrict__ lm,
                                   const uint32_t *__restrict__ seg, const uint32_t *__restrict__ d_n, uint64_t n_cap, uint32_t n_seg,
                                   uint32_t n_lm, const double *__restrict__ knots, const uint32_t *__restrict__ knot_off,
                                   const uint32_t *__restrict__ cp_off, ResRecord *__restrict__ rec, uint32_t *__restrict__ flags) {
    const uint64_t n = d_n ? (uint64_t) *d_n : n_cap;
    const uint64_t k = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0 && n > n_cap) atomicOr(flags, 8u);
    if (k >= n || k >= n_cap) return;
    ResRecord r;
    r.u = obs[2 * k];
    r.v = obs[2 * k + 1];
    r.t = time[k];
    r.lm = lm[k];
    r.seg = seg ? seg[k] : 0u;
    uint32_t f = 0;
    if (r.seg >= n_seg) {
        f |= 4u;
    } else {
        const double *kn = knots + knot_off[r.seg];
        const uint32_t ncp = cp_off[r.seg + 1] - cp_off[r.seg];
        if (r.t < kn[3] || r.t > kn[ncp]) f |= 1u;
        if (k > 0) {
            const uint32_t sp = seg ? seg[k - 1] : 0u;
            if (sp > r.seg) f |= 4u;
            if (sp == r.seg && r.t < time[k - 1]) f |= 1u;
        }
    }
    if (r.lm >= n_lm) f |= 2u;
    if (f) atomicOr(flags, f);
    rec[k] = r;
}

// one thread per (segment, span): the span's residual range [a, b) — [u_span, u_span+1) by lower bounds on the time, the first
// span from the segment's first residual, the last to its last — and how many equal chunks it is cut into (ecal_solver_create)
__global__ void solver_span_kernel(const ResRecord *__restrict__ rec, const uint32_t *__restrict__ d_n, uint64_t n_cap, uint32_t n_seg,
                                   const double *__restrict__ knots, const uint32_t *__restrict__ knot_off,
                                   const uint32_t *__restrict__ cp_off, const uint32_t *__restrict__ span_off /*[n_seg + 1]*/,
                                   uint32_t *__restrict__ span_a, uint32_t *__restrict__ span_m, uint32_t *__restrict__ span_chunks) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= span_off[n_seg]) return;
    const uint32_t n = (uint32_t) (d_n ? (uint64_t) *d_n : n_cap);
    uint32_t g = 0;
    while (j >= span_off[g + 1]) g++;
    const uint32_t ncp = cp_off[g + 1] - cp_off[g], span = 3u + (j - span_off[g]);
    const double *kn = knots + knot_off[g];
    // first index whose (segment, time) is not below (sg, tv)
    auto lower = [&](uint32_t sg, double tv, bool any_time) {
        uint32_t lo = 0, hi = n;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            const ResRecord &r = rec[mid];
            const bool below = r.seg < sg || (r.seg == sg && !any_time && r.t < tv);
            if (below) lo = mid + 1;
            else hi = mid;
        }
        return lo;
    };
    const uint32_t a = span == 3u ? lower(g, 0.0, true) : lower(g, kn[span], false);
    const uint32_t b = span == ncp - 1u ? lower(g + 1u, 0.0, true) : lower(g, kn[span + 1u], false);
    const uint32_t m = b > a ? b - a : 0u;
    span_a[j] = a;
    span_m[j] = m;
    uint32_t nch = 0;
    if (m) {
        const uint32_t parts = (m + NE_CHUNK - 1u) / NE_CHUNK;
        const uint32_t per = ((m + parts - 1u) / parts + NE_T - 1u) / NE_T * NE_T;
        nch = (m + per - 1u) / per;
    }
    span_chunks[j] = nch;
}

// exclusive scan of the spans' chunk counts (one workgroup), then every span writes its chunks
__global__ __launch_bounds__(1024) void solver_chunks_kernel(uint32_t n_spans, uint32_t n_seg, const uint32_t *__restrict__ span_off,
                                                             const uint32_t *__restrict__ span_a, const uint32_t *__restrict__ span_m,
                                                             const uint32_t *__restrict__ span_chunks, Chunk *__restrict__ chunks,
                                                             uint32_t chunk_cap, uint32_t *__restrict__ n_chunks_out) {
    __shared__ uint32_t red[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (uint32_t j0 = 0; j0 < n_spans; j0 += 1024) {
        const uint32_t j = j0 + threadIdx.x;
        const uint32_t v = j < n_spans ? span_chunks[j] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        if (lane == 63) red[wave] = inc;
        __syncthreads();
        uint32_t pre = 0, tot = 0;
        for (int w = 0; w < 16; w++) {
            if (w < wave) pre += red[w];
            tot += red[w];
        }
        if (j < n_spans && v) {
            uint32_t at = carry + pre + inc - v, g = 0;
            while (j >= span_off[g + 1]) g++;
            const uint32_t m = span_m[j], a = span_a[j], parts = (m + NE_CHUNK - 1u) / NE_CHUNK;
            const uint32_t per = ((m + parts - 1u) / parts + NE_T - 1u) / NE_T * NE_T;
            for (uint32_t c = 0; c < m; c += per, at++)
                if (at < chunk_cap) chunks[at] = Chunk{a + c, per < m - c ? per : m - c, g, 3u + (j - span_off[g])};
        }
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_chunks_out = carry;
}

}  // namespace

extern "C" int ecal_solver_create_dev(ecal_ctx *ctx, const ecal_spline_problem *p, const uint32_t *d_n_res, void *stream,
                                      ecal_solver **out) {
    if (!ctx || !p || !out) return ECAL_ERR_INVALID;
    *out = nullptr;
    if (p->n_segments < 1 || !p->seg_cp_off || !p->knots || !p->landmarks || (p->n_res && (!p->obs || !p->time || !p->lm_id))) {
        ctx->last_error = "null pointer / no segment";
        return ECAL_ERR_INVALID;
    }
    if (p->n_res > 0xFFFFFFFFull) return ECAL_ERR_RANGE;
    ecal_solver *s = new (std::nothrow) ecal_solver;
    if (!s) return ECAL_ERR_NOMEM;
    s->ctx = ctx;
    s->n_seg = p->n_segments;
    s->radius = p->circle_radius;
    s->huber_a = p->huber_a;
    s->use_so3 = p->use_so3 != 0;
    s->fisheye = p->camera_model == ECAL_CAMERA_FISHEYE;
    s->cp_off.assign(p->seg_cp_off, p->seg_cp_off + p->n_segments + 1);
    s->n_cp = s->cp_off[p->n_segments];
    s->knot_off.resize(p->n_segments + 1);
    for (uint32_t g = 0; g <= p->n_segments; g++) s->knot_off[g] = s->cp_off[g] + 4 * g;
    s->knots.assign(p->knots, p->knots + s->knot_off[p->n_segments]);
    std::vector<uint32_t> span_off(p->n_segments + 1, 0u);
    for (uint32_t g = 0; g < p->n_segments; g++) {
        if (s->cp_off[g + 1] - s->cp_off[g] < 4) {
            ctx->last_error = "a spline segment needs at least degree + 1 = 4 control points";
            delete s;
            return ECAL_ERR_INVALID;
        }
        span_off[g + 1] = span_off[g] + (s->cp_off[g + 1] - s->cp_off[g] - 3u);
    }
    const uint32_t n_spans = span_off[p->n_segments];
    const size_t n_cap = p->n_res;
    const size_t chunk_cap = n_cap / NE_CHUNK + n_spans + 1;   // a span of m residuals is cut into <= m / NE_CHUNK + 1 chunks
    hipStream_t st = (hipStream_t) stream;
    hipError_t e = hipSetDevice(ctx->device);
    uint32_t *d_tmp = nullptr;   // span_off [n_seg + 1] | span_a, span_m, span_chunks [n_spans] each | flags, n_chunks
    const size_t tmp_words = (p->n_segments + 1) + 3 * (size_t) n_spans + 4;
    auto up = [&](void **dst, const void *src, size_t bytes) {
        if (e != hipSuccess) return;
        e = hipMalloc(dst, bytes ? bytes : 16);
        if (e == hipSuccess && bytes) e = hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, st);
    };
    if (e == hipSuccess) e = hipMalloc((void **) &s->d_rec, (n_cap ? n_cap : 1) * sizeof(ResRecord));
    if (e == hipSuccess) e = hipMalloc((void **) &s->d_chunks, chunk_cap * sizeof(Chunk));
    if (e == hipSuccess) e = hipMalloc((void **) &d_tmp, tmp_words * sizeof(uint32_t));
    up((void **) &s->d_knots, s->knots.data(), s->knots.size() * sizeof(double));
    up((void **) &s->d_landmarks, p->landmarks, 3 * (size_t) p->n_landmarks * sizeof(double));
    up((void **) &s->d_knot_off, s->knot_off.data(), s->knot_off.size() * sizeof(uint32_t));
    up((void **) &s->d_cp_off, s->cp_off.data(), s->cp_off.size() * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **) &s->d_params, s->n_params() * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **) &s->d_accum, s->n_accum() * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **) &s->d_heads, NE_REPL * ACC_HEAD * sizeof(double));
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&normal_eq_kernel<false, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int) (NE_T * NE_LD * sizeof(double)));
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&normal_eq_kernel<true, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int) (NE_T * NE_LD * sizeof(double)));
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&normal_eq_kernel<false, true, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int) (NE_T * NE_LD * sizeof(double)));
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&normal_eq_kernel<true, true, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int) (NE_T * NE_LD * sizeof(double)));
    uint32_t h[2] = {0, 0}, h_n = (uint32_t) n_cap;
    if (e == hipSuccess) {
        uint32_t *d_span_off = d_tmp, *d_a = d_span_off + p->n_segments + 1, *d_m = d_a + n_spans, *d_c = d_m + n_spans,
                 *d_flags = d_c + n_spans;
        e = hipMemcpyAsync(d_span_off, span_off.data(), span_off.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemsetAsync(d_flags, 0, 4 * sizeof(uint32_t), st);
        if (e == hipSuccess && n_cap) {
            hipLaunchKernelGGL(solver_pack_kernel, dim3((unsigned) ((n_cap + 255) / 256)), dim3(256), 0, st, p->obs, p->time, p->lm_id, p->seg_id,
                               d_n_res, (uint64_t) n_cap, p->n_segments, p->n_landmarks, (const double *) s->d_knots,
                               (const uint32_t *) s->d_knot_off, (const uint32_t *) s->d_cp_off, s->d_rec, d_flags);
            hipLaunchKernelGGL(solver_span_kernel, dim3((n_spans + 255) / 256), dim3(256), 0, st, (const ResRecord *) s->d_rec, d_n_res,
                               (uint64_t) n_cap, p->n_segments, (const double *) s->d_knots, (const uint32_t *) s->d_knot_off,
                               (const uint32_t *) s->d_cp_off, (const uint32_t *) d_span_off, d_a, d_m, d_c);
            hipLaunchKernelGGL(solver_chunks_kernel, dim3(1), dim3(1024), 0, st, n_spans, p->n_segments, (const uint32_t *) d_span_off,
                               (const uint32_t *) d_a, (const uint32_t *) d_m, (const uint32_t *) d_c, s->d_chunks, (uint32_t) chunk_cap,
                               d_flags + 1);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(h, d_flags, sizeof(h), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && d_n_res) e = hipMemcpyAsync(&h_n, d_n_res, sizeof(uint32_t), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);   // (the one round trip of the call: 12 bytes)
    }
    if (d_tmp) (void) hipFree(d_tmp);
    if (e != hipSuccess) {
        ctx->last_error = std::string("ecal_solver_create_dev: ") + hipGetErrorString(e);
        ecal_solver_destroy(s);
        return e == hipErrorOutOfMemory ? ECAL_ERR_NOMEM : ECAL_ERR_HIP;
    }
    if (h[0] || h[1] > chunk_cap) {
        ctx->last_error = (h[0] & 8u)   ? "more residuals than the arrays hold (*d_n_res > problem.n_res)"
                          : (h[0] & 4u) ? "seg_id must be non-decreasing and < n_segments"
                          : (h[0] & 2u) ? "landmark id out of range"
                                        : "residual times must be sorted inside a segment and lie inside its knot range";
        ecal_solver_destroy(s);
        return (h[0] & 8u) ? ECAL_ERR_RANGE : ECAL_ERR_INVALID;
    }
    s->n_res = h_n;
    s->n_chunks = h[1];
    *out = s;
    return ECAL_OK;
}

extern "C" uint64_t ecal_solver_num_residuals(const ecal_solver *s) { return s ? s->n_res : 0; }

static int solver_evaluate_dev(ecal_solver *s, const double *d_params, int with_jacobian, double *d_accum, void *stream,
                               const NeProgress *d_prog, uint32_t epoch);
extern "C" int ecal_solver_evaluate_dev(ecal_solver *s, const double *d_params, int with_jacobian, double *d_accum,
                                        void *stream) {
    return solver_evaluate_dev(s, d_params, with_jacobian, d_accum, stream, nullptr, 0);
}
// d_prog (with_jacobian only): the streamed form — the records reach the host's pinned buffer group by group (NeProgress)
static int solver_evaluate_dev(ecal_solver *s, const double *d_params, int with_jacobian, double *d_accum, void *stream,
                               const NeProgress *d_prog, uint32_t epoch) {
    const ecal_range range__(s ? s->ctx : nullptr, with_jacobian ? "ecal_solver_evaluate (normal equations)" : "ecal_solver_evaluate (cost)");
    if (!s || !d_params || !d_accum) return ECAL_ERR_INVALID;
    ecal_ctx *ctx = s->ctx;
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t) stream;
    ECAL_HIP_TRY(ctx, hipMemsetAsync(d_accum, 0, (with_jacobian ? s->n_accum() : 1) * sizeof(double), st));
    ECAL_HIP_TRY(ctx, hipMemsetAsync(s->d_heads, 0, NE_REPL * ACC_HEAD * sizeof(double), st));
    if (s->n_chunks) {
        const size_t lds = with_jacobian ? NE_T * NE_LD * sizeof(double) : 0;
#define ECAL_NE_LAUNCH(SO3_, JAC_, FISH_)                                                                                          \
    hipLaunchKernelGGL((normal_eq_kernel<SO3_, JAC_, FISH_>), dim3(s->n_chunks), dim3(NE_T), lds, st, s->d_rec, s->d_chunks, s->d_knots, \
                       s->d_knot_off, s->d_cp_off, d_params, s->n_cp, s->d_landmarks, s->radius, s->huber_a, d_accum, s->d_heads,   \
                       with_jacobian ? d_prog : nullptr, epoch)
#define ECAL_NE_LAUNCH2(SO3_, JAC_)                                       \
    do {                                                                  \
        if (s->fisheye) ECAL_NE_LAUNCH(SO3_, JAC_, true);                 \
        else ECAL_NE_LAUNCH(SO3_, JAC_, false);                           \
    } while (0)
        if (s->use_so3) {
            if (with_jacobian) ECAL_NE_LAUNCH2(true, true); else ECAL_NE_LAUNCH2(true, false);
        } else {
            if (with_jacobian) ECAL_NE_LAUNCH2(false, true); else ECAL_NE_LAUNCH2(false, false);
        }
#undef ECAL_NE_LAUNCH2
#undef ECAL_NE_LAUNCH
        hipLaunchKernelGGL(reduce_heads_kernel, dim3(1), dim3(128), 0, st, s->d_heads, d_accum,
                           with_jacobian ? (uint32_t) ACC_HEAD : 1u);
        ECAL_HIP_TRY(ctx, hipGetLastError());
    }
    return ECAL_OK;
}

extern "C" int ecal_residuals_dev(ecal_solver *s, const double *d_params, double *d_r, double *d_J, uint32_t *d_cp0, void *stream) {
    if (!s || !d_params || (s->n_res && !d_r)) return ECAL_ERR_INVALID;
    ecal_ctx *ctx = s->ctx;
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t) stream;
    if (s->n_chunks) {
#define ECAL_RR_LAUNCH(SO3_, FISH_)                                                                                                  \
    hipLaunchKernelGGL((residual_rows_kernel<SO3_, FISH_>), dim3(s->n_chunks), dim3(NE_T), 0, st, s->d_rec, s->d_chunks, s->d_knots, \
                       s->d_knot_off, s->d_cp_off, d_params, s->n_cp, s->d_landmarks, s->radius, d_r, d_J, d_cp0)
        if (s->use_so3) {
            if (s->fisheye) ECAL_RR_LAUNCH(true, true); else ECAL_RR_LAUNCH(true, false);
        } else {
            if (s->fisheye) ECAL_RR_LAUNCH(false, true); else ECAL_RR_LAUNCH(false, false);
        }
#undef ECAL_RR_LAUNCH
        ECAL_HIP_TRY(ctx, hipGetLastError());
    }
    return ECAL_OK;
}

extern "C" int ecal_residuals(ecal_solver *s, const double *params, double *r, double *J, uint32_t *cp0) {
    if (!s || !params || (s->n_res && !r)) return ECAL_ERR_INVALID;
    ecal_ctx *ctx = s->ctx;
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t n = s->n_res;
    double *d_r = nullptr, *d_J = nullptr;
    uint32_t *d_c = nullptr;
    struct Guard {
        void *a = nullptr, *b = nullptr, *c = nullptr;
        ~Guard() {
            if (a) (void) hipFree(a);
            if (b) (void) hipFree(b);
            if (c) (void) hipFree(c);
        }
    } g;
    ECAL_HIP_TRY(ctx, hipMalloc((void **) &d_r, (n ? n : 1) * sizeof(double)));
    g.a = d_r;
    if (J) {
        ECAL_HIP_TRY(ctx, hipMalloc((void **) &d_J, (n ? n : 1) * RES_NJ * sizeof(double)));
        g.b = d_J;
    }
    if (cp0) {
        ECAL_HIP_TRY(ctx, hipMalloc((void **) &d_c, (n ? n : 1) * sizeof(uint32_t)));
        g.c = d_c;
    }
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(s->d_params, params, s->n_params() * sizeof(double), hipMemcpyHostToDevice, st));
    int rc = ecal_residuals_dev(s, s->d_params, d_r, d_J, d_c, st);
    if (rc) return rc;
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(r, d_r, n * sizeof(double), hipMemcpyDeviceToHost, st));
    if (J) ECAL_HIP_TRY(ctx, hipMemcpyAsync(J, d_J, n * RES_NJ * sizeof(double), hipMemcpyDeviceToHost, st));
    if (cp0) ECAL_HIP_TRY(ctx, hipMemcpyAsync(cp0, d_c, n * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    ECAL_HIP_TRY(ctx, hipStreamSynchronize(st));
    return ECAL_OK;
}

extern "C" int ecal_solver_evaluate(ecal_solver *s, const double *params, int with_jacobian, double *accum) {
    if (!s || !params || !accum) return ECAL_ERR_INVALID;
    ecal_ctx *ctx = s->ctx;
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(s->d_params, params, s->n_params() * sizeof(double), hipMemcpyHostToDevice, st));
    int rc = ecal_solver_evaluate_dev(s, s->d_params, with_jacobian, s->d_accum, st);
    if (rc) return rc;
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(accum, s->d_accum, (with_jacobian ? s->n_accum() : 1) * sizeof(double),
                                     hipMemcpyDeviceToHost, st));
    ECAL_HIP_TRY(ctx, hipStreamSynchronize(st));
    return ECAL_OK;
}

// ------------------------------------------------------------------------------------------------
// Host side: block-banded arrow system and Levenberg-Marquardt (Ceres 1.x trust-region loop restated)
// ------------------------------------------------------------------------------------------------
namespace {

#include "arrow_host.hpp"

// x (+) delta: intrinsics and translations add, quaternions take exp(delta) (x) q
void plus(const double *x, const std::vector<double> &d, uint32_t n_cp, bool so3, double *out, uint32_t c_lo = 0, uint32_t c_hi = 0xFFFFFFFFu) {
    const size_t nc = 6 * (size_t) n_cp;
    if (c_lo == 0)
        for (int i = 0; i < 9; i++) out[i] = x[i] + d[nc + i];
    for (uint32_t c = c_lo; c < std::min(c_hi, n_cp); c++) {
        if (so3)
            so3_plus(x + 9 + 4 * (size_t) c, &d[6 * (size_t) c], out + 9 + 4 * (size_t) c);
        else
            quaternion_plus(x + 9 + 4 * (size_t) c, &d[6 * (size_t) c], out + 9 + 4 * (size_t) c);
        for (int k = 0; k < 3; k++)
            out[9 + 4 * (size_t) n_cp + 3 * (size_t) c + k] = x[9 + 4 * (size_t) n_cp + 3 * (size_t) c + k] + d[6 * (size_t) c + 3 + k];
    }
}

}  // namespace

extern "C" void ecal_lm_default_options(ecal_lm_options *o) {
    if (!o) return;
    o->max_num_iterations = 50;             // Ceres default, not overridden at EventCalibSpline.cpp:238-243
    o->function_tolerance = 1e-10;          // Sophus::Constants<double>::epsilon(), EventCalibSpline.cpp:240
    o->gradient_tolerance = 1e-10;          // :239
    o->parameter_tolerance = 1e-8;          // Ceres default
    o->initial_trust_region_radius = 1e4;   // Ceres defaults below
    o->max_trust_region_radius = 1e16;
    o->min_relative_decrease = 1e-3;
    o->min_lm_diagonal = 1e-6;
    o->max_lm_diagonal = 1e32;
    o->jacobi_scaling = 1;
    o->allreduce = nullptr;
    o->allreduce_user = nullptr;
    o->distributed = 0;
    o->rank = 0;
    o->world_size = 1;
}

extern "C" int ecal_solver_solve(ecal_solver *s, double *params, const ecal_lm_options *opt_in,
                                 ecal_lm_summary *sum) {
    const ecal_range range__(s ? s->ctx : nullptr, "ecal_solver_solve");
    if (!s || !params) return ECAL_ERR_INVALID;
    ecal_lm_options opt;
    if (opt_in) opt = *opt_in; else ecal_lm_default_options(&opt);
    ecal_ctx *ctx = s->ctx;
    // a NULL all-reduce is a rank-local solve, whether or not the context has joined a communicator: collectives are the
    // caller's explicit choice (opt.allreduce = ecal_comm_allreduce, opt.allreduce_user = ctx, rank / world_size set)
    if (opt.allreduce == ecal_comm_allreduce && (opt.allreduce_user != ctx || opt.rank != ctx->comm_rank || opt.world_size != ctx->comm_size)) {
        ctx->last_error = "ecal_solver_solve: ecal_comm_allreduce needs allreduce_user = the solver's context and its rank / world_size";
        return ECAL_ERR_INVALID;
    }
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t np = s->n_params(), na = s->n_accum(), nc = 6 * (size_t) s->n_cp, nt = nc + 9;
    std::vector<double> x(params, params + np), xc(np), delta, scale(nt, 1.0), dd(nt);
    ArrowSystem A;
    ArrowWorkspace ws;
    const auto t_begin = std::chrono::steady_clock::now();
    // pinned staging (the solver keeps it): the 3 MB buffer comes back every evaluation
    if (!s->h_acc) ECAL_HIP_TRY(ctx, hipHostMalloc((void **) &s->h_acc, na * sizeof(double), hipHostMallocDefault));
    if (!s->h_x) ECAL_HIP_TRY(ctx, hipHostMalloc((void **) &s->h_x, np * sizeof(double), hipHostMallocDefault));
    double *const acc = s->h_acc, *const xpin = s->h_x;
    double t_eval = 0, t_lin = 0;
    // Distributed segments (ecal_lm_options.distributed): every rank owns its own spline segments in its own
    // ecal_solver; only the 9 intrinsics are shared.  Per evaluation the 91-double head is all-reduced; per linear solve
    // the 10 x 10 Schur sums (+ a failure flag and one slot per rank for the gradient max-norm); per step four scalars.
    // distributed == 2, time shards of ONE spline: every rank holds the whole spline layout and the residuals
    // of its time range — the spline's control points cut into `world` interiors with 3-control-point separators
    // (arrow_partition; ecal_solver_time_shard_cuts gives the caller the cut times).  Per evaluation the head and the
    // separators' records are all-reduced (91 + 612 (N - 1) doubles), per linear solve the interiors' 46 x 46 Gram blocks
    // (1082 N doubles); every rank factorises its own interior, solves the small reduced system redundantly and
    // back-substitutes its own control points.  Nothing proportional to the number of control points crosses the links
    // until the solution is put together at the end (one all-reduce of the parameter vector).
    const bool ts_mode = opt.distributed == 2 && opt.allreduce != nullptr;
    const bool dist_mode = (opt.distributed == 1 && opt.allreduce != nullptr) || ts_mode;
    const int world = dist_mode ? std::max(1, opt.world_size) : 1, my_rank = dist_mode ? opt.rank : 0;
    if (dist_mode && (my_rank < 0 || my_rank >= world || world > 1024)) return ECAL_ERR_INVALID;
    std::vector<uint32_t> ts_first, ts_num;   // interiors (control points) of the time shards
    if (ts_mode) {
        if (s->n_seg != 1 || (uint32_t) (7 * world) > s->n_cp) {
            ctx->last_error = "time-sharded solve: one spline segment with at least 7 control points per rank";
            return ECAL_ERR_INVALID;
        }
        arrow_partition(s->n_cp, world, ts_first, ts_num);
    }
    constexpr size_t TS_G = (size_t) APZ * (APZ + 1) / 2 + 1;   // upper triangle of an interior's Gram block + its flag
    double *d_small = nullptr, *h_small = nullptr;
    const size_t n_small = std::max<size_t>(128 + (size_t) world, ts_mode ? std::max<size_t>(TS_G * (size_t) world, ACC_HEAD + 3 * ACC_PER_CP * (size_t) (world - 1)) : 0);
    if (dist_mode) {
        ECAL_HIP_TRY(ctx, hipMalloc((void **) &d_small, n_small * sizeof(double)));
        if (hipHostMalloc((void **) &h_small, n_small * sizeof(double), hipHostMallocDefault) != hipSuccess) {
            (void) hipFree(d_small);
            return ECAL_ERR_NOMEM;
        }
    }
    struct FreeSmall {
        double *d, *h;
        ~FreeSmall() {
            if (d) (void) hipFree(d);
            if (h) (void) hipHostFree(h);
        }
    } free_small{d_small, h_small};
    auto reduce_small = [&](double *v, size_t n) -> bool {  // sum v[0..n) over ranks in place
        memcpy(h_small, v, n * sizeof(double));
        if (hipMemcpyAsync(d_small, h_small, n * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess) return false;
        if (opt.allreduce(opt.allreduce_user, d_small, n, st) != 0) return false;
        if (hipMemcpyAsync(h_small, d_small, n * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess) return false;
        if (hipStreamSynchronize(st) != hipSuccess) return false;
        memcpy(v, h_small, n * sizeof(double));
        return true;
    };
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double>(b - a).count();
    };

    auto evaluate = [&](const double *xp, int with_jac, double *cost) -> int {
        const auto te = now();
        memcpy(xpin, xp, np * sizeof(double));
        hipError_t e = hipMemcpyAsync(s->d_params, xpin, np * sizeof(double), hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return ECAL_ERR_HIP;
        int rc = ecal_solver_evaluate_dev(s, s->d_params, with_jac, s->d_accum, st);
        if (rc) return rc;
        const size_t n = with_jac ? na : 1;
        if (opt.allreduce && ts_mode && with_jac) {
            // time shards: the head and the separators' records (3 control points per cut) are what two ranks both add to
            size_t at = ACC_HEAD;
            hipError_t e2 = hipMemcpyAsync(d_small, s->d_accum, ACC_HEAD * sizeof(double), hipMemcpyDeviceToDevice, st);
            for (int c = 0; c + 1 < world && e2 == hipSuccess; c++, at += 3 * ACC_PER_CP)
                e2 = hipMemcpyAsync(d_small + at, s->d_accum + ACC_HEAD + ACC_PER_CP * (size_t) (ts_first[c] + ts_num[c]),
                                    3 * ACC_PER_CP * sizeof(double), hipMemcpyDeviceToDevice, st);
            if (e2 != hipSuccess) return ECAL_ERR_HIP;
            if (opt.allreduce(opt.allreduce_user, d_small, at, st) != 0) return ECAL_ERR_HIP;
            e2 = hipMemcpyAsync(s->d_accum, d_small, ACC_HEAD * sizeof(double), hipMemcpyDeviceToDevice, st);
            at = ACC_HEAD;
            for (int c = 0; c + 1 < world && e2 == hipSuccess; c++, at += 3 * ACC_PER_CP)
                e2 = hipMemcpyAsync(s->d_accum + ACC_HEAD + ACC_PER_CP * (size_t) (ts_first[c] + ts_num[c]), d_small + at,
                                    3 * ACC_PER_CP * sizeof(double), hipMemcpyDeviceToDevice, st);
            if (e2 != hipSuccess) return ECAL_ERR_HIP;
        } else if (opt.allreduce) {  // per-GPU partials summed over ranks (RCCL all-reduce supplied by the caller)
            // distributed segments: only the head (cost, intrinsics gradient and block) is shared between ranks
            rc = opt.allreduce(opt.allreduce_user, s->d_accum, dist_mode ? std::min(n, (size_t) ACC_HEAD) : n, st);
            if (rc) return ECAL_ERR_HIP;
        }
        e = hipMemcpyAsync(acc, s->d_accum, n * sizeof(double), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            ctx->last_error = std::string("solver evaluate: ") + hipGetErrorString(e);
            return ECAL_ERR_HIP;
        }
        *cost = acc[0];
        t_eval += secs(te, now());
        return ECAL_OK;
    };

    // one rank, a long spline: the factorisation, the unpacking and the quadratic forms run on several host cores
    // (arrow_host_parts.hpp); the sharded modes keep the sequential routines (their segments are short, and the sequential
    // routine's 10 x 10 Schur sums are what the ranks exchange)
    ArrowParts parts;
    int n_parts = arrow_parts_for(s->n_cp);
    if ((uint32_t) (7 * n_parts) > s->n_cp) n_parts = 1;
    // (sharded segments: unpacking and the quadratic forms still use the pool, the factorisation stays the sequential routine)
    const bool parts_solve = !dist_mode;
    // time shards: the interiors' Gram blocks (upper triangles) + a "positive definite" flag, one slot per rank, summed
    ArrowParts ts_parts;
    const std::function<bool(ArrowParts &)> ts_exchange = [&](ArrowParts &pt) -> bool {
        std::vector<double> buf(TS_G * (size_t) world, 0.0);
        double *mine = buf.data() + TS_G * (size_t) my_rank;
        const double *G = pt.G.data() + (size_t) my_rank * APZ * APZ;
        size_t k = 0;
        bool finite = true;
        for (int i = 0; i < APZ; i++)
            for (int j = i; j < APZ; j++) {
                finite = finite && std::isfinite(G[(size_t) i * APZ + j]);
                mine[k++] = G[(size_t) i * APZ + j];
            }
        if (!finite)
            for (size_t q = 0; q + 1 < TS_G; q++) mine[q] = 0.0;
        mine[TS_G - 1] = (pt.ok[my_rank] && finite) ? 1.0 : 0.0;
        if (!reduce_small(buf.data(), buf.size())) return false;
        for (int p = 0; p < world; p++) {
            const double *src = buf.data() + TS_G * (size_t) p;
            double *Gp = pt.G.data() + (size_t) p * APZ * APZ;
            size_t q = 0;
            for (int i = 0; i < APZ; i++)
                for (int j = i; j < APZ; j++) Gp[(size_t) i * APZ + j] = src[q++];
            pt.ok[p] = src[TS_G - 1] == 1.0 ? 1 : 0;
        }
        return true;
    };
    const uint32_t epoch_at_start = s->stream_epoch;
    HostPool *pool = nullptr;   // (the solver keeps the threads: starting fifteen of them costs as much as a tenth of an iteration each)
    double t_unpack = 0, t_pool = 0;
    if (n_parts > 1) {
        const auto tp = now();
        const int hw = host_usable_cpus();   // (affinity ∩ cgroup quota ÷ the node's ranks — not hardware_concurrency())
        const int workers = std::max(0, std::min(n_parts, hw) - 1);
        if (!s->host_pool || s->host_pool_workers != workers) {
            try {
                s->host_pool = std::shared_ptr<void>(new HostPool(workers), [](void *q) { delete static_cast<HostPool *>(q); });
                s->host_pool_workers = workers;
            } catch (...) {   // no threads to be had: the sequential routines
                s->host_pool.reset();
                s->host_pool_workers = -1;
                n_parts = 1;
            }
        }
        pool = static_cast<HostPool *>(s->host_pool.get());
        t_pool = secs(tp, now());
    }
    auto unpack_acc = [&]() {
        const auto tu = now();
        if (pool) {   // band rows by ranges of control points, one range per task
            unpack_alloc(s->n_cp, A);
            unpack_head(acc, A);
            const uint32_t T = 4u * (uint32_t) n_parts, per = (s->n_cp + T - 1) / T;
            pool->run((int) T, [&](int t) { unpack_rows(acc, A, std::min(s->n_cp, (uint32_t) t * per), std::min(s->n_cp, ((uint32_t) t + 1) * per)); });
        } else {
            unpack(acc, s->n_cp, A);
        }
        t_unpack += secs(tu, now());
    };

    // Streamed evaluation (one rank, the multi-part host solve): NeProgress above — the kernel delivers the records group by
    // group, the pool's threads unpack an interior's rows and, when the trust-region radius the next linear solve will use
    // can be predicted, factorise it while the kernel is still busy with the later control points.  What is left behind the
    // kernel: the last interior, the separators' rows, the reduced system, the back-substitution.
    bool stream_ok = !opt.allreduce && pool && n_parts > 1 && parts_solve && n_parts <= NE_MAX_GROUPS && !ctx->sw.solver_no_stream &&
                     s->n_chunks > 0;
    std::vector<uint32_t> part_first, part_num;
    if (stream_ok) {
        arrow_partition_stream(s->n_cp, n_parts, part_first, part_num);
        if (s->prog.n_groups != (uint32_t) n_parts) {   // first solve with this partition: the groups' chunk counts, the buffers
            std::vector<Chunk> ch(s->n_chunks);
            ECAL_HIP_TRY(ctx, hipMemcpy(ch.data(), s->d_chunks, ch.size() * sizeof(Chunk), hipMemcpyDeviceToHost));
            NeProgress &pg = s->prog;
            memset(&pg, 0, sizeof(pg));
            pg.n_cp = s->n_cp;
            for (int g = 0; g < n_parts; g++) pg.cut[g] = part_first[g];
            pg.cut[n_parts] = s->n_cp;
            for (const Chunk &c : ch) {
                const uint32_t sg = s->cp_off[c.seg] + c.span;
                uint32_t g = 0;
                while (g + 1 < (uint32_t) n_parts && pg.cut[g + 1] <= sg) g++;
                pg.init[g]++;
            }
            for (int b = 0; b + 1 < n_parts; b++) pg.init[NE_MAX_GROUPS + b] = (pg.init[b] ? 1u : 0u) + (pg.init[b + 1] ? 1u : 0u);
            if (!s->h_flag) {
                ECAL_HIP_TRY(ctx, hipHostMalloc((void **) &s->h_flag, 2 * NE_MAX_GROUPS * sizeof(uint32_t), hipHostMallocDefault));
                memset(s->h_flag, 0, 2 * NE_MAX_GROUPS * sizeof(uint32_t));
            }
            if (!s->d_left) ECAL_HIP_TRY(ctx, hipMalloc((void **) &s->d_left, 2 * NE_MAX_GROUPS * sizeof(uint32_t)));
            if (!s->d_prog) ECAL_HIP_TRY(ctx, hipMalloc((void **) &s->d_prog, sizeof(NeProgress)));
            ECAL_HIP_TRY(ctx, hipStreamSynchronize(st));
            ECAL_HIP_TRY(ctx, hipMemcpy(s->d_left, pg.init, 2 * NE_MAX_GROUPS * sizeof(uint32_t), hipMemcpyHostToDevice));
            pg.left = s->d_left;
            ECAL_HIP_TRY(ctx, hipHostGetDevicePointer((void **) &pg.host_acc, s->h_acc, 0));
            ECAL_HIP_TRY(ctx, hipHostGetDevicePointer((void **) &pg.host_flag, s->h_flag, 0));
            ECAL_HIP_TRY(ctx, hipMemcpy(s->d_prog, &pg, sizeof(pg), hipMemcpyHostToDevice));
            pg.n_groups = (uint32_t) n_parts;
            ECAL_HIP_TRY(ctx, hipMemcpy(s->d_prog, &pg, sizeof(pg), hipMemcpyHostToDevice));
        } else {
            // the kernel restores the groups' counters itself as it finishes each group — unless an earlier evaluation on this
            // solver stopped half way (an error, a stream that failed to deliver): a solve starts from the full counts whatever
            // the last one left behind (128 words; nothing of this solver is in flight here)
            ECAL_HIP_TRY(ctx, hipMemcpyAsync(s->d_left, s->prog.init, 2 * NE_MAX_GROUPS * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        }
    }
    ArrowSystem A_next;
    std::vector<double> dd_next(nt);
    bool reduced_ok = false;   // a streamed evaluation with `factor`: every separator of the reduced system eliminated
    std::vector<double> tl_arrive, tl_done;   // ECAL_TRACE=solver: the last streamed evaluation's timeline (seconds from its start)
    double tl_run = 0, tl_sync = 0;
    double t_tail = 0;
    // An: the system at xp.  factor: also arrow_part_factor of every interior, with the LM diagonal of trust-region radius
    // r_fact (ws / parts then hold what arrow_parts_finish needs).  *streamed = false: the stream failed to deliver (nothing this
    // code can name should make it) and the buffer was fetched and unpacked the plain way, nothing factorised.
    auto evaluate_streamed = [&](const double *xp, ArrowSystem &An, bool factor, double r_fact, double *cost, bool *streamed,
                                 const std::function<void()> *after_launch = nullptr) -> int {
        const auto te = now();
        const NeProgress &pg = s->prog;
        const int P = n_parts;
        memcpy(xpin, xp, np * sizeof(double));
        if (hipMemcpyAsync(s->d_params, xpin, np * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess) return ECAL_ERR_HIP;
        const uint32_t epoch = ++s->stream_epoch;
        // records that no chunk touches never arrive: they are zero
        for (int g = 0; g < P; g++) {
            if (!pg.init[g]) memset(acc + ACC_HEAD + ACC_PER_CP * (size_t) pg.cut[g], 0,
                                    ACC_PER_CP * (size_t) ((g + 1 < P ? pg.cut[g + 1] - 3 : pg.cut[g + 1]) - pg.cut[g]) * sizeof(double));
            if (g + 1 < P && !pg.init[NE_MAX_GROUPS + g]) memset(acc + ACC_HEAD + ACC_PER_CP * (size_t) (pg.cut[g + 1] - 3), 0, 3 * ACC_PER_CP * sizeof(double));
        }
        int rc2 = solver_evaluate_dev(s, s->d_params, 1, s->d_accum, st, s->d_prog, epoch);
        if (rc2) return rc2;
        if (hipMemcpyAsync(acc, s->d_accum, ACC_HEAD * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess) return ECAL_ERR_HIP;
        if (after_launch) (*after_launch)();   // (host work of the caller that only has to be done by the time the kernel is)
        unpack_alloc(s->n_cp, An);
        if (factor) arrow_parts_setup(An.nc, P, ws, parts, true);
        reduced_ok = false;
        const bool trace_ev = ctx->sw.solver_trace;
        tl_arrive.assign(P + 1, 0.0);
        tl_done.assign(P + 1, 0.0);
        // the host tasks (arrow_host.hpp: an interior per task as its records arrive, the separators behind them)
        StreamedSource src;
        src.init = pg.init;
        src.cut = pg.cut;
        src.flag = s->h_flag;
        src.epoch = epoch;
        src.producer_gone = [&]() -> bool { return hipStreamQuery(st) != hipErrorNotReady; };
        const bool delivered = arrow_streamed_tasks(pool, P, src, acc, An, factor, r_fact, scale.data(), dd_next.data(), opt.min_lm_diagonal,
                                                    opt.max_lm_diagonal, ws, parts, &reduced_ok, trace_ev ? tl_arrive.data() : nullptr,
                                                    trace_ev ? tl_done.data() : nullptr, te);
        tl_run = secs(te, now());
        const hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            ctx->last_error = std::string("solver evaluate: ") + hipGetErrorString(e);
            return ECAL_ERR_HIP;
        }
        const auto tt = now();
        *streamed = delivered;
        if (!delivered) {
            if (hipMemcpy(acc, s->d_accum, na * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return ECAL_ERR_HIP;
            unpack(acc, s->n_cp, An);
        } else {
            unpack_head(acc, An);
            if (!factor)
                for (int p = 0; p + 1 < P; p++) unpack_rows(acc, An, pg.cut[p + 1] - 3, pg.cut[p + 1]);
        }
        *cost = acc[0];
        tl_sync = secs(te, tt);
        t_tail += secs(tt, now());
        t_eval += secs(te, now());
        return ECAL_OK;
    };

    ecal_lm_summary S;
    memset(&S, 0, sizeof(S));
    double cost = 0, radius = opt.initial_trust_region_radius, decrease_factor = 2.0;
    int rc;
    bool fact_ready = false;      // ws / parts hold the interiors' factors for A with the diagonal of radius fact_radius
    double fact_radius = 0;
    int n_prefactored = 0;
    if (stream_ok) {
        bool streamed = false;
        rc = evaluate_streamed(x.data(), A, false, 0.0, &cost, &streamed, nullptr);   // (the column scaling comes from this evaluation: nothing to factorise with yet)
        if (rc) return rc;
        if (!streamed) stream_ok = false;
    } else {
        rc = evaluate(x.data(), 1, &cost);
        if (rc) return rc;
        unpack_acc();
    }
    S.jacobian_evaluations = 1;
    S.initial_cost = cost;
    if (opt.jacobi_scaling) {  // computed once from the initial Jacobian, as Ceres does
        for (size_t i = 0; i < nc; i++) scale[i] = 1.0 / (1.0 + std::sqrt(A.band[i * BW]));
        for (int i = 0; i < 9; i++) scale[nc + i] = 1.0 / (1.0 + std::sqrt(A.corner[10 * i]));
    }
    S.termination = 1;  // NO_CONVERGENCE unless a test fires
    auto gmax = [&]() {
        double m = 0;
        for (size_t i = 0; i < nc; i++) m = std::max(m, std::fabs(A.gc[i]));
        if (dist_mode) {  // one slot per rank, summed: a max over ranks without a max collective
            std::vector<double> v((size_t) world, 0.0);
            v[(size_t) my_rank] = m;
            if (reduce_small(v.data(), v.size()))
                for (double x : v) m = std::max(m, x);
        }
        for (int i = 0; i < 9; i++) m = std::max(m, std::fabs(A.gi[i]));
        return m;
    };
    if (dist_mode)
        ws.reduce_G = [&](double *G) -> bool {
            double buf[101];
            bool bad = false;
            for (int i = 0; i < 100; i++) {
                buf[i] = G[i];
                bad = bad || !std::isfinite(G[i]);
            }
            buf[100] = bad ? 1.0 : 0.0;
            if (bad)
                for (int i = 0; i < 100; i++) buf[i] = 0.0;
            if (!reduce_small(buf, 101)) return false;
            for (int i = 0; i < 100; i++) G[i] = buf[i];
            return buf[100] == 0.0;
        };
    if (gmax() <= opt.gradient_tolerance) S.termination = 0;
    bool last_step_ok = true;
    // the interiors factorised and the separators eliminated while the kernel ran: the intrinsics' corner and the way back
    double t_fin_end = 0, t_fin_back = 0;
    auto prefactored_finish = [&]() -> bool {
        const auto ta = now();
        if (!reduced_ok || !arrow_reduced_end(A, scale.data(), dd.data(), parts)) return false;
        const auto tb = now();
        arrow_parts_backsub(A.nc, delta, ws, parts, pool, n_parts);
        t_fin_end += secs(ta, tb);
        t_fin_back += secs(tb, now());
        return true;
    };
    double t_dd = 0, t_fin = 0, t_quad = 0, t_plus = 0, t_book = 0;   // ECAL_TRACE=solver: the host's share of an iteration, by item
    while (S.termination == 1 && S.iterations < opt.max_num_iterations) {
        S.iterations++;
        const auto t_it = now();
        // Levenberg-Marquardt diagonal on the scaled system (the streamed evaluation left the control points' part behind when
        // its radius is the one in force)
        const bool prefactored = fact_ready && fact_radius == radius && reduced_ok;
        if (prefactored) memcpy(dd.data(), dd_next.data(), nc * sizeof(double));
        for (size_t i = prefactored ? nc : 0; i < nt; i++) {
            const double h = (i < nc ? A.band[i * BW] : A.corner[10 * (i - nc)]) * scale[i] * scale[i];
            dd[i] = std::min(std::max(h, opt.min_lm_diagonal), opt.max_lm_diagonal) / radius;
        }
        const auto tl = now();
        t_dd += secs(t_it, tl);
        bool ok = ts_mode ? solve_arrow_parts(A, scale, dd, delta, ws, ts_parts, nullptr, world, my_rank, &ts_exchange)
                  : (n_parts > 1 && parts_solve)
                      ? (fact_ready && fact_radius == radius ? prefactored_finish()   // (false when an interior or a separator was not positive definite)
                                                             : solve_arrow_parts(A, scale, dd, delta, ws, parts, pool, n_parts))   // (a whole factorisation: the even partition — the streamed evaluation sets up its own)
                      : solve_arrow(A, scale, dd, delta, ws);
        if (fact_ready && fact_radius == radius) n_prefactored++;
        fact_ready = false;
        const auto t_q = now();
        t_fin += secs(tl, t_q);
        double model_change = 0;
        // After a successful step the next one is usually successful too: evaluate the candidate WITH its normal
        // equations in one pass (4.8 ms) instead of a cost-only pass (1.0 ms + a host round trip) followed, on
        // acceptance, by the full pass at the same point.  After a rejected step fall back to the cost-only probe.
        const bool speculate = last_step_ok;
        // streamed evaluation ahead: the quadratic forms of the model are computed once the kernel is running (they gate the
        // evaluation only when the step is no descent step of the model, which a positive definite system rules out up to rounding)
        const bool defer_quad = ok && !dist_mode && speculate && stream_ok;
        double gTd_late = 0, dHd_late = 0;
        if (defer_quad) {
            for (size_t i = 0; i < nt; i++) delta[i] *= scale[i];
            model_change = 1.0;   // (placeholder until the forms are in)
        } else if (ok && ts_mode) {
            // the step solves (H + D) y = -g exactly, so y^T H y = -g^T y - y^T D y and the model change -g^T y - y^T H y / 2 is
            // (y^T D y - g^T y) / 2: sums over unknowns — this rank's interior, rank 0 also the separators and the intrinsics
            // (every rank holds the same values for those)
            double two[2] = {0, 0};
            auto add = [&](size_t i) {
                const double g = i < nc ? A.gc[i] : A.gi[i - nc];
                two[0] += g * scale[i] * delta[i];
                two[1] += dd[i] * delta[i] * delta[i];
            };
            for (size_t i = 6 * (size_t) ts_first[my_rank]; i < 6 * (size_t) (ts_first[my_rank] + ts_num[my_rank]); i++) add(i);
            if (my_rank == 0) {
                for (int c = 0; c + 1 < world; c++)
                    for (size_t i = 6 * (size_t) (ts_first[c] + ts_num[c]); i < 6 * (size_t) (ts_first[c] + ts_num[c] + 3); i++) add(i);
                for (size_t i = nc; i < nt; i++) add(i);
            }
            if (!reduce_small(two, 2)) return ECAL_ERR_HIP;
            for (size_t i = 0; i < nt; i++) delta[i] *= scale[i];
            model_change = 0.5 * (two[1] - two[0]);
            ok = model_change > 0.0;
        } else if (ok) {
            for (size_t i = 0; i < nt; i++) delta[i] *= scale[i];
            double gTd, dHd;
            quad_forms(A, delta, &gTd, &dHd, dist_mode && my_rank != 0, pool, n_parts);
            if (dist_mode) {
                double two[2] = {gTd, dHd};
                if (!reduce_small(two, 2)) return ECAL_ERR_HIP;
                gTd = two[0];
                dHd = two[1];
            }
            model_change = -gTd - 0.5 * dHd;
            ok = model_change > 0.0;
        }  // (a failed factorisation was agreed on through reduce_G: every rank skips the reduction above together)
        t_lin += secs(tl, now());
        t_quad += secs(t_q, now());
        if (!ok) {  // invalid step: shrink the region
            radius /= decrease_factor;
            decrease_factor *= 2.0;
            S.unsuccessful_steps++;
            continue;
        }
        const auto t_p = now();
        plus(x.data(), delta, s->n_cp, s->use_so3, xc.data());   // (on the pool's threads: measured slower, 47 against 28 us)
        t_plus += secs(t_p, now());
        double new_cost;
        // streamed: the interiors are factorised for the radius a step with rel >= 0.937 leads to (the usual one while the
        // model is good: radius / max(1/3, 1 - (2 rel - 1)^3) = 3 radius); any other verdict factorises again as before
        const double r_pred = std::min(opt.max_trust_region_radius, radius / std::max(1.0 / 3.0, 0.0));
        bool cand_in_next = false, cand_factored = false;   // the candidate's system sits unpacked in A_next / its interiors are factorised
        if (speculate && stream_ok) {
            const std::function<void()> late = [&] { quad_forms(A, delta, &gTd_late, &dHd_late, false, pool, n_parts); };
            rc = evaluate_streamed(xc.data(), A_next, true, r_pred, &new_cost, &cand_factored, defer_quad ? &late : nullptr);
            cand_in_next = true;
            if (!rc && !cand_factored) stream_ok = false;   // (fetched the plain way: carry on without the stream)
        } else {
            rc = evaluate(xc.data(), speculate ? 1 : 0, &new_cost);
        }
        if (rc) return rc;
        const auto t_b = now();
        if (defer_quad) {
            model_change = -gTd_late - 0.5 * dHd_late;
            if (!(model_change > 0.0)) {   // invalid step after all: the evaluation is dropped, the region shrinks
                S.jacobian_evaluations++;
                radius /= decrease_factor;
                decrease_factor *= 2.0;
                S.unsuccessful_steps++;
                continue;
            }
        }
        if (speculate) S.jacobian_evaluations++;
        else S.cost_evaluations++;
        const double rel = (cost - new_cost) / model_change;
        double step2 = 0, x2 = 0;
        if (ts_mode) {   // own interior from every rank; separators and intrinsics once (rank 0)
            auto own = [&](uint32_t c) {
                for (int k = 0; k < 6; k++) step2 += delta[6 * (size_t) c + k] * delta[6 * (size_t) c + k];
                for (int k = 0; k < 4; k++) x2 += x[9 + 4 * (size_t) c + k] * x[9 + 4 * (size_t) c + k];
                for (int k = 0; k < 3; k++) x2 += x[9 + 4 * (size_t) s->n_cp + 3 * (size_t) c + k] * x[9 + 4 * (size_t) s->n_cp + 3 * (size_t) c + k];
            };
            for (uint32_t c = ts_first[my_rank]; c < ts_first[my_rank] + ts_num[my_rank]; c++) own(c);
            if (my_rank == 0) {
                for (int q = 0; q + 1 < world; q++)
                    for (uint32_t c = ts_first[q] + ts_num[q]; c < ts_first[q] + ts_num[q] + 3; c++) own(c);
                for (size_t i = nc; i < nt; i++) step2 += delta[i] * delta[i];
                for (size_t i = 0; i < 9; i++) x2 += x[i] * x[i];
            }
            double two[2] = {step2, x2};
            if (!reduce_small(two, 2)) return ECAL_ERR_HIP;
            step2 = two[0];
            x2 = two[1];
        } else if (dist_mode) {  // own control points from every rank, the shared intrinsics once
            for (size_t i = 0; i < nc; i++) step2 += delta[i] * delta[i];
            for (size_t i = 9; i < np; i++) x2 += x[i] * x[i];
            double two[2] = {step2, x2};
            if (!reduce_small(two, 2)) return ECAL_ERR_HIP;
            step2 = two[0];
            x2 = two[1];
            for (size_t i = nc; i < nt; i++) step2 += delta[i] * delta[i];
            for (size_t i = 0; i < 9; i++) x2 += x[i] * x[i];
        } else {
            for (size_t i = 0; i < nt; i++) step2 += delta[i] * delta[i];
            for (size_t i = 0; i < np; i++) x2 += x[i] * x[i];
        }
        if (rel > opt.min_relative_decrease) {
            const double cost_change = cost - new_cost;
            x.swap(xc);
            const double prev = cost;
            if (speculate) {
                cost = new_cost;  // the buffer of the speculative pass is the one to unpack
            } else {
                rc = evaluate(x.data(), 1, &cost);
                if (rc) return rc;
                S.jacobian_evaluations++;
            }
            if (cand_in_next) std::swap(A, A_next);   // (unpacked while the kernel ran)
            else unpack_acc();
            S.successful_steps++;
            last_step_ok = true;
            const double t = 2.0 * rel - 1.0;
            radius = std::min(opt.max_trust_region_radius, radius / std::max(1.0 / 3.0, 1.0 - t * t * t));
            if (cand_factored) {
                fact_ready = true;
                fact_radius = r_pred;
            }
            decrease_factor = 2.0;
            if (gmax() <= opt.gradient_tolerance) S.termination = 0;
            else if (std::fabs(cost_change) <= opt.function_tolerance * prev) S.termination = 0;
        } else {
            radius /= decrease_factor;
            decrease_factor *= 2.0;
            S.unsuccessful_steps++;
            last_step_ok = false;
        }
        if (S.termination == 1 && std::sqrt(step2) <= opt.parameter_tolerance * (std::sqrt(x2) + opt.parameter_tolerance))
            S.termination = 0;
        t_book += secs(t_b, now());
    }
    if (ts_mode) {
        // the solution put together: every rank contributes its interior, rank 0 the separators and the intrinsics (the one
        // exchange proportional to the spline's length, once per solve)
        std::vector<double> mine(np, 0.0);
        auto take = [&](uint32_t c) {
            for (int k = 0; k < 4; k++) mine[9 + 4 * (size_t) c + k] = x[9 + 4 * (size_t) c + k];
            for (int k = 0; k < 3; k++) mine[9 + 4 * (size_t) s->n_cp + 3 * (size_t) c + k] = x[9 + 4 * (size_t) s->n_cp + 3 * (size_t) c + k];
        };
        for (uint32_t c = ts_first[my_rank]; c < ts_first[my_rank] + ts_num[my_rank]; c++) take(c);
        if (my_rank == 0) {
            for (int q = 0; q + 1 < world; q++)
                for (uint32_t c = ts_first[q] + ts_num[q]; c < ts_first[q] + ts_num[q] + 3; c++) take(c);
            for (int i = 0; i < 9; i++) mine[i] = x[i];
        }
        memcpy(xpin, mine.data(), np * sizeof(double));
        if (hipMemcpyAsync(s->d_params, xpin, np * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess) return ECAL_ERR_HIP;
        if (opt.allreduce(opt.allreduce_user, s->d_params, np, st) != 0) return ECAL_ERR_HIP;
        if (hipMemcpyAsync(xpin, s->d_params, np * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess) return ECAL_ERR_HIP;
        if (hipStreamSynchronize(st) != hipSuccess) return ECAL_ERR_HIP;
        memcpy(x.data(), xpin, np * sizeof(double));
    }
    S.final_cost = cost;
    S.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    S.seconds_evaluate = t_eval;
    S.seconds_linear_solve = t_lin;
    s->last_solve[0] = (uint32_t) n_parts;
    s->last_solve[1] = s->stream_epoch - epoch_at_start;
    s->last_solve[2] = (uint32_t) n_prefactored;
    s->last_solve[3] = pool ? (uint32_t) pool->workers() : 0u;
    s->last_solve[4] = stream_ok ? 1u : 0u;
    s->last_solve[5] = (uint32_t) S.iterations;
    s->last_solve[6] = (uint32_t) dist_mode | (ts_mode ? 2u : 0u);
    if (s->ctx->sw.solver_trace)
        fprintf(stderr, "ecal_solver_solve: total %.4f s | evaluate %.4f | linear solve %.4f (%d parts) | unpack %.4f | pool %.4f | streamed evaluations: "
                        "%u, behind the kernel %.4f, %d of %d linear solves found their interiors factorised\n", S.seconds,
                t_eval, t_lin, n_parts, t_unpack, t_pool, s->stream_epoch, t_tail, n_prefactored, S.iterations);
    if (s->ctx->sw.solver_trace)
        fprintf(stderr, "  host items, ms over the solve: LM diagonal %.3f | factorise / finish %.3f | scaling + quadratic forms %.3f | plus %.3f | "
                        "verdict, norms, swap / unpack, gradient norm %.3f | of the finishes: corner %.3f, back-substitution %.3f\n", 1e3 * t_dd, 1e3 * t_fin,
                1e3 * t_quad, 1e3 * t_plus, 1e3 * t_book, 1e3 * t_fin_end, 1e3 * t_fin_back);
    if (s->ctx->sw.solver_trace && !tl_arrive.empty()) {
        fprintf(stderr, "  last streamed evaluation, ms from its start: interiors arrived / factorised");
        for (size_t p = 0; p + 1 < tl_arrive.size(); p++) fprintf(stderr, " %.2f/%.2f", 1e3 * tl_arrive[p], 1e3 * tl_done[p]);
        fprintf(stderr, " | separators done %.2f | tasks joined %.2f | stream synchronised %.2f\n", 1e3 * tl_done.back(), 1e3 * tl_run, 1e3 * tl_sync);
    }
    memcpy(params, x.data(), np * sizeof(double));
    if (sum) *sum = S;
    return ECAL_OK;
}

// The time shards of one spline (ecal_lm_options.distributed == 2): cut_time[c] for c < world - 1 — rank r owns the residuals
// with cut_time[r - 1] <= t < cut_time[r] (rank 0 from the first knot, the last rank to the last).  The cuts sit on knots: the
// control points are cut into `world` interiors with 3-control-point separators, the partition of the host's own multi-core
// solve (arrow_partition); a residual left of the knot behind a separator touches no control point right of it and vice versa.
extern "C" int ecal_solver_time_shard_cuts(const double *knots, uint32_t n_cp, int world, double *cut_time) {
    if (!knots || !cut_time || world < 1 || (uint32_t) (7 * world) > n_cp) return ECAL_ERR_INVALID;
    std::vector<uint32_t> f, m;
    arrow_partition(n_cp, world, f, m);
    for (int c = 0; c + 1 < world; c++) cut_time[c] = knots[f[c] + m[c] + 3];   // spans up to (first separator CP + 2) stay left
    return ECAL_OK;
}

// closed form of PinholeCamera::inverseRadialDistortion (core/sensor/src/PinholeCamera.cpp:69-95)
extern "C" void ecal_inverse_radial_distortion(const double *k4, double *b5) {
    const double k1 = k4[0], k2 = k4[1], k3 = k4[2], k4v = k4[3];
    const double k1_2 = k1 * k1, k1_3 = k1_2 * k1, k1_4 = k1_3 * k1, k1_5 = k1_4 * k1;
    b5[0] = -k1;
    b5[1] = 3 * k1_2 - k2;
    b5[2] = -12 * k1_3 + 8 * k1 * k2 - k3;
    b5[3] = 55 * k1_4 - 55 * k1_2 * k2 + 5 * k2 * k2 + 10 * k1 * k3 - k4v;
    b5[4] = -273 * k1_5 + 364 * k1_3 * k2 - 78 * k1 * k2 * k2 - 78 * k1_2 * k3 + 12 * k2 * k3 + 12 * k1 * k4v;
}

// Test hook: one linear solve (S A S + D) y = -S g, delta = S y, for the normal equations `accum` (host, ecal_solver_normal_size
// doubles) and the column scaling `scale` (host, 6 n_cp + 9) by the host routines (sequential, partitioned, the streamed
// evaluation's partition); tests/test_gpu_solver.py requires them to agree.
// The host's linear solves alone (no GPU involved): the accumulation buffer of n_cp control points (ecal_solver_normal_size
// layout) -> the LM step.  mode 0: the sequential routine (solve_arrow); 2: the partitioned one (arrow_host_parts.hpp) with `parts`
// interiors on a few threads; 3: the same on the streamed evaluation's partition (interiors shrinking towards the end).
extern "C" int ecal_debug_arrow_solve_host(uint32_t n_cp, const double *accum, const double *scale, double radius, double min_diag,
                                           double max_diag, double *delta_out, int *fail_out, int mode, int parts_wanted) {
    const int rc = arrow_debug_solve_host(n_cp, accum, scale, radius, min_diag, max_diag, delta_out, fail_out, mode, parts_wanted);
    return rc == -1 ? ECAL_ERR_INVALID : (rc == -6 ? ECAL_ERR_RANGE : ECAL_OK);
}

// tests: how the last ecal_solver_solve on this solver ran — [0] interiors of the host's partition, [1] streamed evaluations,
// [2] linear solves that found their interiors factorised, [3] worker threads of the pool, [4] 1 = the streamed path was in force
// at the end, [5] iterations, [6] bit 0 distributed, bit 1 time-sharded
extern "C" int ecal_debug_solver_last_solve(const ecal_solver *s, uint32_t *out8) {
    if (!s || !out8) return ECAL_ERR_INVALID;
    memcpy(out8, s->last_solve, sizeof(s->last_solve));
    return ECAL_OK;
}

// tests / bench: the CPUs the solver's worker pool is sized for (affinity ∩ cgroup quota ÷ LOCAL_WORLD_SIZE; ECAL_HOST_THREADS
// overrides) and, in *node_quota, the figure before the division by the node's ranks.  No GPU involved.
extern "C" int ecal_debug_host_usable_cpus(int *node_quota) { return host_usable_cpus(node_quota); }

// tests (no GPU involved): the solve's worker pool — `rounds` runs of 1 .. 40 tasks on `workers` threads, with pauses long enough for
// the workers to go to sleep now and then, nudges from inside tasks (as the streamed evaluation's last interior does) and from the
// caller; every task of every run must have run exactly once, in a run of its own.  Returns the number of violations.
extern "C" int ecal_debug_host_pool_selftest(int workers, int rounds) { return host_pool_selftest(workers, rounds); }

// tests: the host solve's partitions (arrow_partition / arrow_partition_stream)
extern "C" int ecal_debug_arrow_partition(uint32_t n_cp, int parts, int stream, uint32_t *first_cp, uint32_t *num_cp) {
    if (!first_cp || !num_cp || parts < 1 || (uint32_t) (7 * parts) > n_cp) return ECAL_ERR_INVALID;
    std::vector<uint32_t> f, m;
    if (stream) arrow_partition_stream(n_cp, parts, f, m);
    else arrow_partition(n_cp, parts, f, m);
    memcpy(first_cp, f.data(), parts * sizeof(uint32_t));
    memcpy(num_cp, m.data(), parts * sizeof(uint32_t));
    return ECAL_OK;
}

extern "C" int ecal_debug_arrow_solve(ecal_solver *s, const double *accum, const double *scale, double radius, double min_diag,
                                      double max_diag, double *delta_out, int *fail_out, int mode) {
    if (!s || !accum || !scale || !delta_out || !fail_out || mode == 1) return ECAL_ERR_INVALID;   // (1 was the device form, round 3 - 5)
    return ecal_debug_arrow_solve_host(s->n_cp, accum, scale, radius, min_diag, max_diag, delta_out, fail_out, mode, 0);
}

