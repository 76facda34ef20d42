// Ordering of the candidate circles into the asymmetric pattern grid.
//
// Replaces what the reference gets from its vendored, patched OpenCV finder —
// cv::findCirclesGrid(points, Size(cols, rows), centers, CALIB_CB_ASYMMETRIC_GRID [| CLUSTERING]) at
// event_camera_calib/src/CirclesEventFrame.cpp:332-336 (cv_calib/src/cv_calib.cpp:7-87,
// cv_calib/src/circlesgrid.cpp) followed by the nearest-candidate lookup of :340-353 — with a
// deterministic lattice walk (the reference's finder runs kmeans with random centres, x100 attempts).
// Output convention = the reference's (circlesgrid.cpp:1258-1291): grid index i*cols + j is the model
// point ((2j + i%2) s, i s) (EventCalibIni.cpp:102-106).  For rows x cols = 9 x 4 the pattern has no
// orientation-preserving self-symmetry, so a complete grid has exactly one valid assignment; the walk
// assumes the board is seen from its front (model -> image mapping with positive determinant).
// Parity with OpenCV's finder is NOT pinned (third-party algorithm, random by construction); tests check
// the ordering against synthetic ground truth.
//
// One wave (64 lanes) per window: lanes share the nearest-candidate searches, lane-uniform control flow
// does the breadth-first walk over the centred-square lattice (neighbours along the two diagonals).
#include "ecal_ctx.hpp"
#include "row_direction.hpp"

#pragma clang fp contract(off)

namespace ecal {

constexpr int GR_T = 64;
constexpr uint32_t GR_MAXC = 128;  // candidates per window handled
constexpr int GR_L = 32;           // lattice coordinate window (u, v in [-16, 15])
constexpr uint32_t GR_MAXM = 128;  // pattern points handled (rows * cols)
constexpr uint32_t GR_PARALLEL_MAX = 1024;   // windows at work up to which the launch takes the latency form (a wave per start)

// the changes of lattice basis the pattern is matched through: the four rotations, then the other integer matrices
// (a b; c d) of determinant + 1 with entries in [-2, 2], in ascending order of (a + 2) + 5 (b + 2) + 25 (c + 2) + 125 (d + 2)
constexpr uint32_t GR_NTF = 52;
__constant__ int8_t GR_TF[GR_NTF][4] = {{1, 0, 0, 1}, {0, -1, 1, 0}, {-1, 0, 0, -1}, {0, 1, -1, 0}, {-1, -1, -1, -2}, {0, 1, -1, -2}, {0, -1, 1, -2}, {-1, 1, 1, -2}, {-1, 0, -2, -1}, {1, 1, -2, -1}, {-2, -1, -1, -1}, {-1, 0, -1, -1}, {0, 1, -1, -1}, {1, 2, -1, -1}, {-1, -2, 0, -1}, {-1, -1, 0, -1}, {-1, 1, 0, -1}, {-1, 2, 0, -1}, {1, -2, 1, -1}, {0, -1, 1, -1}, {-1, 0, 1, -1}, {-2, 1, 1, -1}, {1, -1, 2, -1}, {-1, 0, 2, -1}, {-2, 1, -1, 0}, {-1, 1, -1, 0}, {1, 1, -1, 0}, {2, 1, -1, 0}, {-2, -1, 1, 0}, {-1, -1, 1, 0}, {1, -1, 1, 0}, {2, -1, 1, 0}, {1, 0, -2, 1}, {-1, 1, -2, 1}, {2, -1, -1, 1}, {1, 0, -1, 1}, {0, 1, -1, 1}, {-1, 2, -1, 1}, {1, -2, 0, 1}, {1, -1, 0, 1}, {1, 1, 0, 1}, {1, 2, 0, 1}, {-1, -2, 1, 1}, {0, -1, 1, 1}, {1, 0, 1, 1}, {2, 1, 1, 1}, {-1, -1, 2, 1}, {1, 0, 2, 1}, {1, -1, -1, 2}, {0, 1, -1, 2}, {0, -1, 1, 2}, {1, 1, 1, 2}};

// minimum over the wave, in every lane: a DPP reduction towards lane 63 (row shifts, then the row broadcasts) + a read of that
// lane — vector-ALU moves instead of twelve dependent trips through the LDS crossbar (__shfl_xor), which were most of the
// walk's latency: the walk asks for a nearest candidate ~150 times per window, one after the other
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#define ECAL_GR_MIN_STEP(ctrl, row_mask)                                                                              \
    {                                                                                                                 \
        const uint32_t lo = (uint32_t) __builtin_amdgcn_update_dpp(-1, (int) (uint32_t) v, ctrl, row_mask, 0xF, false);   \
        const uint32_t hi = (uint32_t) __builtin_amdgcn_update_dpp(-1, (int) (uint32_t) (v >> 32), ctrl, row_mask, 0xF, false); \
        const unsigned long long w = ((unsigned long long) hi << 32) | lo;                                             \
        v = w < v ? w : v;                                                                                            \
    }
    ECAL_GR_MIN_STEP(0x111, 0xF)   // row_shr:1
    ECAL_GR_MIN_STEP(0x112, 0xF)   // row_shr:2
    ECAL_GR_MIN_STEP(0x114, 0xF)   // row_shr:4
    ECAL_GR_MIN_STEP(0x118, 0xF)   // row_shr:8: lane 15 of every row holds the row's minimum
    ECAL_GR_MIN_STEP(0x142, 0xA)   // row_bcast:15 -> rows 1, 3
    ECAL_GR_MIN_STEP(0x143, 0xC)   // row_bcast:31 -> rows 2, 3: lane 63 holds the wave's
#undef ECAL_GR_MIN_STEP
    const uint32_t lo = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) v, 63);
    const uint32_t hi = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (v >> 32), 63);
    return ((unsigned long long) hi << 32) | lo;
}

// minimum over a ROW of 16 lanes, in every lane of the row (four rotations): the wave as four independent searchers
__device__ __forceinline__ unsigned long long row_min_u64(unsigned long long v) {
#define ECAL_GR_ROW_STEP(ctrl)                                                                                        \
    {                                                                                                                 \
        const uint32_t lo = (uint32_t) __builtin_amdgcn_update_dpp(-1, (int) (uint32_t) v, ctrl, 0xF, 0xF, false);    \
        const uint32_t hi = (uint32_t) __builtin_amdgcn_update_dpp(-1, (int) (uint32_t) (v >> 32), ctrl, 0xF, 0xF, false); \
        const unsigned long long w = ((unsigned long long) hi << 32) | lo;                                             \
        v = w < v ? w : v;                                                                                            \
    }
    ECAL_GR_ROW_STEP(0x128)   // row_ror:8
    ECAL_GR_ROW_STEP(0x124)   // row_ror:4
    ECAL_GR_ROW_STEP(0x122)   // row_ror:2
    ECAL_GR_ROW_STEP(0x121)   // row_ror:1
#undef ECAL_GR_ROW_STEP
    return v;
}
__device__ __forceinline__ unsigned long long lane_u64(unsigned long long v, int src) {   // lane src's value, in a scalar register pair
    const uint32_t lo = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) v, src);
    const uint32_t hi = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (v >> 32), src);
    return ((unsigned long long) hi << 32) | lo;
}
__device__ __forceinline__ double lane_f64(double v, int src) { return __longlong_as_double((long long) lane_u64((unsigned long long) __double_as_longlong(v), src)); }

__device__ __forceinline__ unsigned long long pack_key(double d2, uint32_t idx) {
    // non-negative doubles order like their bit patterns; low 8 bits carry the index (ties: smaller index)
    return (((unsigned long long) __double_as_longlong(d2)) & ~0xFFull) | (idx & 0xFFu);
}

#ifdef ECAL_PHASE_PROF
static __device__ unsigned long long g_gr_cycles[16];
static __device__ unsigned long long g_gr_out[16];    // [o]: windows by outcome, [8 + o]: their cycles
static __device__ unsigned long long g_gr_hist[32];   // [0..23]: windows by log2 of their cycles; [24]: max (cycles << 24 | n << 16 | nodes << 8 | sweeps)
#define GR_MARK(i)                                                       \
    do {                                                                 \
        if (lane == 0) {                                                 \
            const unsigned long long now__ = __builtin_amdgcn_s_memtime(); \
            atomicAdd(&g_gr_cycles[i], now__ - gr_t__);                  \
            gr_t__ = now__;                                              \
        }                                                                \
    } while (0)
#else
#define GR_MARK(i)
#endif

// NW = 1: one wave per window, the starts one after the other (the throughput form: a pass of thousands of windows).
// NW = 4 (round 5): the LATENCY form for passes of few windows — the keyframe search's tail, where this kernel's time is the time
// of its slowest window, one that goes through every start and fails (1.16 M cycles against 0.3 M for a window found at once).
// The starts are independent of each other (every one wipes the previous walk; which seed a later start takes follows from the
// basis searches of the earlier ones alone), so wave w runs start w with a working set of its own — replaying the earlier starts'
// basis searches for its seed —, a start that finds the grid tells the later ones to stop, and the lowest successful start's
// result is the window's: the same result as the sequential form (tests/test_gpu_grid.py runs the two against each other).
template <int NW>
__global__ __launch_bounds__(GR_T * NW) void grid_order_kernel(const uint32_t *__restrict__ win_info,
                                                          const uint32_t *__restrict__ seg_off,
                                                          const double *__restrict__ cand_xyr, uint32_t rows,
                                                          uint32_t cols, double tol_frac, double tol_px, int32_t *__restrict__ order,
                                                          uint32_t *__restrict__ found, int debug,
                                                          double *__restrict__ dirs /* [S][rows][2] or NULL: see the end */) {
    __shared__ double px_a[NW][GR_MAXC], py_a[NW][GR_MAXC];
    __shared__ double e1x_a[NW][GR_MAXC], e1y_a[NW][GR_MAXC], e2x_a[NW][GR_MAXC], e2y_a[NW][GR_MAXC];  // local lattice basis per node
    __shared__ int8_t cu_a[NW][GR_MAXC], cv_a[NW][GR_MAXC];
    __shared__ uint8_t assigned_a[NW][GR_MAXC], queue_a[NW][GR_MAXC];
    __shared__ uint8_t occ_a[NW][GR_L * GR_L];  // lattice cell -> candidate index + 1
    __shared__ double hm_a[NW][24];             // moment sums of the partial grid's homography fit (second attempt)
    __shared__ double hh_a[NW][8];
    __shared__ uint32_t sh_qt_a[NW];
    __shared__ int sh_box_a[NW][4];                       // box of the visited lattice cells (match_pattern)
    __shared__ int8_t mU_a[NW][GR_MAXM], mV_a[NW][GR_MAXM];     // lattice coordinates of the model points relative to model point 0
    __shared__ int8_t tf_sh_a[NW][4 * GR_NTF];            // the table of basis changes
    __shared__ uint8_t sel_a[NW][GR_MAXM];                // the matched candidate of every model point
    __shared__ uint32_t best_start_sh;                    // NW > 1: the lowest start that has found the grid (255: none yet)
    const uint32_t s = blockIdx.x, lane = threadIdx.x & 63u, wv = NW > 1 ? threadIdx.x >> 6 : 0u;
    double *const px = px_a[wv], *const py = py_a[wv], *const e1x = e1x_a[wv], *const e1y = e1y_a[wv], *const e2x = e2x_a[wv], *const e2y = e2y_a[wv];
    int8_t *const cu = cu_a[wv], *const cv = cv_a[wv], *const mU = mU_a[wv], *const mV = mV_a[wv], *const tf_sh = tf_sh_a[wv];
    uint8_t *const assigned = assigned_a[wv], *const queue = queue_a[wv], *const occ = occ_a[wv], *const sel = sel_a[wv];
    double *const hm = hm_a[wv], *const hh = hh_a[wv];
    uint32_t &sh_qt = sh_qt_a[wv];
    int *const sh_box = sh_box_a[wv];
    // (every wave works on arrays of its own: a wave-level barrier orders its LDS traffic; NW = 1 keeps the workgroup barrier)
#define GR_SYNC()                                                        \
    do {                                                                 \
        if constexpr (NW == 1) {                                         \
            __syncthreads();                                             \
        } else {                                                         \
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");       \
            __builtin_amdgcn_wave_barrier();                             \
        }                                                                \
    } while (0)
    const uint32_t n = win_info[4 * (size_t) s], M = rows * cols;
    int32_t *out = order + (size_t) s * M;
    for (uint32_t m = lane; m < M; m += GR_T) out[m] = -1;
    if (lane == 0) found[s] = 0;
    if (ECAL_WIN_STATUS(win_info[4 * (size_t) s + 3]) != 0 || n < M || n > GR_MAXC || M > GR_MAXM) return;
    const double *c = cand_xyr + 3 * (size_t) seg_off[2 * s];
    for (uint32_t i = lane; i < n; i += GR_T) {
        px[i] = c[3 * i];
        py[i] = c[3 * i + 1];
        assigned[i] = 0;
    }
    for (uint32_t k = lane; k < GR_L * GR_L; k += GR_T) occ[k] = 0;
    for (uint32_t m = lane; m < M; m += GR_T) {   // model point (x, y) = (2 j + i % 2, i): U = (x + y) / 2, V = (y - x) / 2
        const int i = (int) (m / cols), jj = (int) (m % cols);
        const int x = 2 * jj + (i & 1), y = i;
        mU[m] = (int8_t) ((x + y) / 2);
        mV[m] = (int8_t) ((y - x) / 2);
    }
    for (uint32_t k = lane; k < 4u * GR_NTF; k += GR_T) tf_sh[k] = GR_TF[k / 4u][k % 4u];
    if constexpr (NW > 1) {
        if (threadIdx.x == 0) best_start_sh = 255u;
        __syncthreads();   // (every wave of the window is here: the conditions of the returns above are the window's)
    }
    GR_SYNC();

    // The searches of the seed, the basis and the first walk run on REGISTERS: every row of 16 lanes holds all candidates
    // (candidate gl + 16 k in lane gl's k-th register, the four rows alike), so a row answers a nearest-candidate query with
    // eight distance keys per lane and four DPP rotations — no LDS, no barrier — and the wave answers four different queries
    // at once (the four directions of a walk step).  The walk asked ~150 questions one after the other, each a trip through
    // LDS, a six-step wave reduction and a barrier: half of this kernel's 250 us per window (profiles/r04_notes.md).
    const uint32_t gl = lane & 15u, grp = lane >> 4;
    double rx[8], ry[8];
    uint32_t rvalid = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t i = gl + 16u * (uint32_t) k;
        rx[k] = i < n ? px[i] : 0.0;
        ry[k] = i < n ? py[i] : 0.0;
        rvalid |= i < n ? 1u << k : 0u;
    }
    // nearest candidate to (qx, qy) — per row: the rows may ask different questions — among this lane's candidates in `mask`
    // with a key above `floor_key` (the k-th nearest = the nearest above the (k-1)-th's key: keys are unique, the index is in them)
    const int rK = (int) ((n + 15u) / 16u);   // registers in use (the same in every lane)
    auto row_nearest = [&](double qx, double qy, uint32_t mask, bool above, unsigned long long floor_key) __attribute__((always_inline)) -> unsigned long long {
        unsigned long long best = ~0ull;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (k >= rK) break;
            const double dx = rx[k] - qx, dy = ry[k] - qy;
            const unsigned long long key = pack_key(dx * dx + dy * dy, gl + 16u * (uint32_t) k);
            const bool take = ((mask >> k) & 1u) && (!above || key > floor_key) && key < best;
            best = take ? key : best;
        }
        return row_min_u64(best);
    };
    auto bit_of = [&](uint32_t j) -> uint32_t { return (j & 15u) == gl ? 1u << (j >> 4) : 0u; };   // this lane's mask bit of candidate j

    // seed: the candidate closest to the centroid (an interior circle has all four diagonal neighbours)
    double sx = 0, sy = 0;
    for (uint32_t i = lane; i < n; i += GR_T) {
        sx += px[i];
        sy += py[i];
    }
    for (int o = 32; o > 0; o >>= 1) {
        sx += __shfl_xor(sx, o, 64);
        sy += __shfl_xor(sy, o, 64);
    }
#ifdef ECAL_PHASE_PROF
    unsigned long long gr_t__ = __builtin_amdgcn_s_memtime();
    const unsigned long long gr_t0__ = gr_t__;
    uint32_t gr_sweeps__ = 0;
    if (lane == 0) atomicAdd(&g_gr_cycles[15], 1ull);
    uint32_t gr_outcome__ = 4;   // 0 / 1: found by the first start's walk / second attempt; 2 / 3: by a later start's; 4: not found
    auto gr_done__ = [&](uint32_t nodes) {
        if (lane == 0) {
            const unsigned long long cyc = __builtin_amdgcn_s_memtime() - gr_t0__;
            const uint32_t oc = gr_outcome__ == 4u && n == M ? 5u : gr_outcome__;   // 5: not found with exactly M candidates
            atomicAdd(&g_gr_out[oc], 1ull);
            atomicAdd(&g_gr_out[8 + oc], cyc);
            atomicAdd(&g_gr_hist[63 - __clzll((long long) (cyc | 1ull)) < 23 ? 63 - __clzll((long long) (cyc | 1ull)) : 23], 1ull);
            atomicMax(&g_gr_hist[24], (cyc << 24) | ((unsigned long long) (n & 0xFFu) << 16) | ((unsigned long long) (nodes & 0xFFu) << 8) | (gr_sweeps__ & 0xFFu));
        }
    };
#endif
    // Seed and basis, made robust against clutter INSIDE the pattern (round 4; spurious candidates between the circles are what
    // the reference's primary finder absorbs by voting — findBasis clusters ALL the neighbourhood-graph edge vectors,
    // circlesgrid.cpp:978-1044 — and what a seed or a step taken from one bad neighbour does not survive).  A lattice point
    // inside the pattern has its neighbours in antipodal pairs: s + v and s - v are both candidates.  A spurious candidate has
    // no such pairs, and a spurious neighbour of a true point has no antipode.  The FIRST start is the plain one, as it has
    // always been — the candidate nearest the centroid, its nearest neighbour and the nearest one not parallel to it, both
    // attempts —, so nothing that was found before is lost and a window that is found at once costs what it did.  When it finds
    // no grid (round 4: only with more candidates than pattern points; round 6: always), a ROBUST start follows: the seed is the
    // candidate nearest the centroid that has two independent PAIRED neighbour vectors (GR_SEEDS are looked at), its steps the
    // plain rule's when both of those are paired, else the shortest two independent paired vectors; both attempts when the
    // first walk places two thirds of the pattern.  Then up to two more starts from the next such seeds, first walk only.  (Measured on the 50 M-event search, 1270 pieces: five extra starts
    // with second attempts cost 25 % of the search's time; tests/test_gpu_grid.py states the verdicts under clutter.)
    constexpr int GR_NB = 8, GR_SEEDS = 5, GR_EXTRA = 3;
    unsigned long long seeds_p = 0;   // the seeds, a byte each (an array indexed at run time would live in scratch memory)
    auto seeds = [&](int q) -> uint32_t { return (uint32_t) (seeds_p >> (8 * q)) & 0xFFu; };
    int n_seeds = 0;
    {
        unsigned long long floor_key = 0;
        for (int st = 0; st < GR_SEEDS; st++) {   // the candidates nearest the centroid, nearest first
            const unsigned long long rs = lane_u64(row_nearest(sx / n, sy / n, rvalid, st > 0, floor_key), 0);
            if (rs == ~0ull) break;
            seeds_p |= (rs & 0xFFull) << (8 * n_seeds);
            n_seeds++;
            floor_key = rs;
        }
    }
    if (n_seeds == 0) return;
    // the two steps of a walk from `seed` (all candidates free on entry and on return); false: none
    // mode 0: the first start's rule (false: the seed has no two independent paired vectors — try the next seed);
    // mode 1: paired vectors only; mode 2: the plain rule whatever the pairs say
    auto choose_basis = [&](uint32_t seed, int mode, double &ax, double &ay, double &bx, double &by) __attribute__((always_inline)) -> bool {
        unsigned long long nb_p = 0;   // the neighbours, a byte each
        auto nb = [&](int k) -> uint32_t { return (uint32_t) (nb_p >> (8 * k)) & 0xFFu; };
        const uint32_t not_seed = rvalid & ~bit_of(seed);
        {
            unsigned long long floor_key = 0;
            bool none = false;
            for (int k = 0; k < GR_NB; k++) {   // the eight nearest neighbours, nearest first
                const unsigned long long r = none ? ~0ull : lane_u64(row_nearest(px[seed], py[seed], not_seed, k > 0, floor_key), 0);
                none = r == ~0ull;
                nb_p |= (unsigned long long) (none ? seed : (uint32_t) (r & 0xFFu)) << (8 * k);
                floor_key = r;
            }
        }
        // antipodal pairs: neighbour k counts when some other candidate sits at s - (p_k - s), within 0.3 of the step
        // (four neighbours at a time: a row of lanes per question)
        uint32_t paired = 0;
        if (mode != 2)
#pragma unroll
            for (int k0 = 0; k0 < GR_NB; k0 += 4) {
                const uint32_t nbk = nb(k0 + (int) grp);
                const double vxr = px[nbk] - px[seed], vyr = py[nbk] - py[seed];
                const unsigned long long rr = row_nearest(px[seed] - vxr, py[seed] - vyr, not_seed, false, 0);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int k = k0 + q;
                    if (nb((int) (k)) == seed) continue;
                    const unsigned long long r = lane_u64(rr, 16 * q);
                    if (r == ~0ull) continue;
                    const double vx = px[nb((int) (k))] - px[seed], vy = py[nb((int) (k))] - py[seed];
                    const uint32_t j = (uint32_t) (r & 0xFFu);
                    const double ex = px[j] - (px[seed] - vx), ey = py[j] - (py[seed] - vy);
                    if (j != nb((int) (k)) && ex * ex + ey * ey <= 0.09 * (vx * vx + vy * vy)) paired |= 1u << k;
                }
            }
        auto parallel = [&](uint32_t ka, uint32_t kb) -> bool {
            const double ux = px[nb((int) (ka))] - px[seed], uy = py[nb((int) (ka))] - py[seed], dx = px[nb((int) (kb))] - px[seed], dy = py[nb((int) (kb))] - py[seed];
            const double cr = fabs(ux * dy - uy * dx), nn = sqrt((ux * ux + uy * uy) * (dx * dx + dy * dy));
            return !(nn > 0 && cr / nn > 0.5);
        };
        auto short_enough = [&](uint32_t ka, uint32_t kb) -> bool {   // not longer than four times the first
            const double ux = px[nb((int) (ka))] - px[seed], uy = py[nb((int) (ka))] - py[seed], dx = px[nb((int) (kb))] - px[seed], dy = py[nb((int) (kb))] - py[seed];
            return (dx * dx + dy * dy) < 16.0 * (ux * ux + uy * uy);
        };
        // the plain rule: the nearest neighbour and the FIRST one not parallel to it (too long: no lattice here)
        int qa = -1, qb = -1;
        if (nb((int) (0)) != seed) {
            qa = 0;
            for (int k = 1; k < GR_NB; k++) {
                if (nb((int) (k)) == seed) break;
                if (!parallel(0u, (uint32_t) k)) {
                    if (short_enough(0u, (uint32_t) k)) qb = k;
                    break;
                }
            }
        }
        // the paired rule: the shortest two independent paired vectors
        int ka = -1, kb = -1;
        for (int k = 0; k < GR_NB && kb < 0; k++) {
            if (nb((int) (k)) == seed) break;
            if (!((paired >> k) & 1u)) continue;
            if (ka < 0) ka = k;
            else if (!parallel((uint32_t) ka, (uint32_t) k) && short_enough((uint32_t) ka, (uint32_t) k)) kb = k;
        }
        if (mode == 2 || (mode == 0 && qb >= 0 && ((paired >> qa) & 1u) && ((paired >> qb) & 1u))) {
            ka = qa;
            kb = qb;
        }
        if (ka < 0 || kb < 0) return false;
        ax = px[nb((int) (ka))] - px[seed];
        ay = py[nb((int) (ka))] - py[seed];
        bx = px[nb((int) (kb))] - px[seed];
        by = py[nb((int) (kb))] - py[seed];
        if (ax * by - ay * bx < 0) {  // right-handed (u, v) in image coordinates
            bx = -bx;
            by = -by;
        }
        return true;
    };
    bool got = false;
    uint32_t qh = 0, qt = 0;
    // fit_h: the homography lattice (u, v) -> image through the placed nodes (inhomogeneous DLT, 8 x 8 normal equations)
    // (always_inline: a lambda the compiler leaves as a function takes its captures by address — `qt`, `lane`, … would live in
    // scratch memory for the whole kernel, a trip to memory per use inside the walk)
    auto fit_h = [&]() __attribute__((always_inline)) {
        // The normal equations of the node rows [u v 1 0 0 0 -ux -vx | x], [0 0 0 u v 1 -uy -vy | y] are made of 24 sums
        // S(a, b, g) = sum over the nodes of u^a v^b g with a + b <= 2 and g in {1, x, y, x^2 + y^2}: a lane per sum
        // (index 4 * m + g, m = 0..5 for u^a v^b = 1, u, v, u^2, u v, v^2), nodes in queue order.
        if (lane < 24u) {
            const uint32_t m = lane >> 2, g = lane & 3u;
            double acc = 0;
            for (uint32_t k = 0; k < qt; k++) {
                const uint32_t j = queue[k];
                const double u = cu[j], v = cv[j], x = px[j], y = py[j];
                const double mono = m == 0u ? 1.0 : (m == 1u ? u : (m == 2u ? v : (m == 3u ? u * u : (m == 4u ? u * v : v * v))));
                const double gg = g == 0u ? 1.0 : (g == 1u ? x : (g == 2u ? y : x * x + y * y));
                acc += mono * gg;
            }
            hm[lane] = acc;
        }
        GR_SYNC();
        // row r of the 8 x 9 system in lane r's registers.  S1(p, q) = sum of t_p t_q g for t = (u, v, 1)
        auto S = [&](uint32_t p, uint32_t q, uint32_t g) -> double {   // p, q in {0: u, 1: v, 2: 1}
            const uint32_t lo = p < q ? p : q, hi = p < q ? q : p;
            // (u,u) 3  (u,v) 4  (u,1) 1  (v,v) 5  (v,1) 2  (1,1) 0
            const uint32_t m = lo == 0u ? (hi == 0u ? 3u : (hi == 1u ? 4u : 1u)) : (lo == 1u ? (hi == 1u ? 5u : 2u) : 0u);
            return hm[4u * m + g];
        };
        double row[9];
        {
            const uint32_t r = lane & 7u;
#pragma unroll
            for (uint32_t c = 0; c < 9; c++) {
                double val;
                if (r < 3u) {          // d/dh of the x rows: t_r * [t | 0 | -t_{0,1} x | x]
                    val = c < 3u ? S(r, c, 0) : (c < 6u ? 0.0 : (c < 8u ? -S(r, c - 6u, 1) : S(r, 2, 1)));
                } else if (r < 6u) {   // the y rows
                    val = c < 3u ? 0.0 : (c < 6u ? S(r - 3u, c - 3u, 0) : (c < 8u ? -S(r - 3u, c - 6u, 2) : S(r - 3u, 2, 2)));
                } else {               // -t_{r-6} x * x row + -t_{r-6} y * y row
                    val = c < 3u ? -S(r - 6u, c, 1) : (c < 6u ? -S(r - 6u, c - 3u, 2) : (c < 8u ? S(r - 6u, c - 6u, 3) : -S(r - 6u, 2, 3)));
                }
                row[c] = val;
            }
        }
        // Gaussian elimination with partial pivoting on the eight lanes' registers: the pivot row of a column is the unused
        // row with the largest entry (the first such in row order), broadcast with v_readlane; no row is moved
        auto bcast = [](double v, uint32_t src) -> double {
            const long long b = __double_as_longlong(v);
            const uint32_t lo = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) b, (int) src);
            const uint32_t hi = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (b >> 32), (int) src);
            return __longlong_as_double((long long) (((unsigned long long) hi << 32) | lo));
        };
        uint32_t used = 0, perm[8];
        bool okh = true;
#pragma unroll
        for (int c = 0; c < 8; c++) {
            uint32_t piv = 0xFFFFFFFFu;
            double best = -1.0;
#pragma unroll
            for (uint32_t r = 0; r < 8; r++) {
                const double v = fabs(bcast(row[c], r));
                const bool take = !((used >> r) & 1u) && v > best;
                best = take ? v : best;
                piv = take ? r : piv;
            }
            if (!(best > 1e-12)) okh = false;
            piv = okh ? piv : 0u;
            perm[c] = piv;
            used |= 1u << piv;
            const double pc = bcast(row[c], piv);
            const bool mine = lane < 8u && !((used >> lane) & 1u);
            const double f = row[c] / pc;
#pragma unroll
            for (int k = c; k < 9; k++) {
                const double pk = bcast(row[k], piv);
                row[k] = mine ? row[k] - f * pk : row[k];
            }
        }
        double x[8];
#pragma unroll
        for (int c = 7; c >= 0; c--) {
            double t = row[8];
#pragma unroll
            for (int k = c + 1; k < 8; k++) t -= row[k] * x[k];
            x[c] = bcast(t / row[c], perm[c]);
        }
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < 8; c++) hh[c] = okh ? x[c] : NAN;
        }
        GR_SYNC();
    };
    auto fit_all = [&]() __attribute__((always_inline)) { fit_h(); };
    int seed_at = 0;   // the next seed to look at
    // NW > 1: this wave's start alone — after the basis searches of the starts before it, which say where its seed search
    // begins (a start without a basis ends the sequence: then this wave has no start to run)
    bool runnable = true;
    if constexpr (NW > 1) {
        if (wv > 0u && n < M) runnable = false;
        double t0_, t1_, t2_, t3_;
        if (runnable && wv >= 2u) {   // start 1's seed search
            bool have1 = false;
            for (int q = 0; q < n_seeds && !have1; q++) {
                have1 = choose_basis(seeds(q), 0, t0_, t1_, t2_, t3_);
                seed_at = q + 1;
            }
            runnable = have1;
        }
        if (runnable && wv >= 3u) {   // start 2's
            bool have2 = false;
            while (seed_at < n_seeds && !have2) have2 = choose_basis(seeds(seed_at++), 1, t0_, t1_, t2_, t3_);
            runnable = have2;
        }
    }
    // a later start stops when an earlier one has found the grid (its result could not be the window's)
#define GR_CANCELLED(start_) (NW > 1 && *reinterpret_cast<volatile uint32_t *>(&best_start_sh) < (uint32_t) (start_))
#pragma nounroll
    for (int start = (NW > 1 ? (int) wv : 0); start <= (NW > 1 ? (int) wv : GR_EXTRA) && !got && runnable; start++) {
    // (round 6: the later starts run for windows with EXACTLY the pattern's count of candidates too — until then "no clutter to have
    // misled the first start" ended them here; the midpoint circles of a 4 - 6 step window are up to 8 px off their centres, and the
    // plain start lost 9 of 723 such windows of tests/test_gpu_oracle_chain.py's search that hold the 36 circles and nothing else:
    // 2 of 725 now, the 1270-piece search's time unchanged)
    if (start > 0 && n < M) break;
    if (start > 0) {   // the previous start's walk is wiped
        for (uint32_t i = lane; i < n; i += GR_T) assigned[i] = 0;
        for (uint32_t k = lane; k < GR_L * GR_L; k += GR_T) occ[k] = 0;
        GR_SYNC();
    }
    uint32_t seed = seeds(0);
    double ax = 0, ay = 0, bx = 0, by = 0;
    bool have = false;
    if (start == 0) {          // the plain start, as it has always been: nothing that was found before round 4 is lost
        have = choose_basis(seed, 2, ax, ay, bx, by);
    } else if (start == 1) {   // the robust start: the first seed with two independent paired vectors, plain steps if THEY are paired
        for (int q = 0; q < n_seeds && !have; q++) {
            seed = seeds(q);
            have = choose_basis(seed, 0, ax, ay, bx, by);
            seed_at = q + 1;
        }
    } else {                   // the next seeds, paired steps
        while (seed_at < n_seeds && !have) {
            seed = seeds(seed_at++);
            have = choose_basis(seed, 1, ax, ay, bx, by);
        }
    }
    if (!have) {
        if ((debug & 1) && lane == 0) out[8 + start] = -1000;
        if (start == 0) continue;
        break;
    }
    GR_MARK(0);   // seed + basis
    // breadth-first walk
    qh = 0;
    qt = 0;
    if (lane == 0) {
        assigned[seed] = 1;
        cu[seed] = 0;
        cv[seed] = 0;
        e1x[seed] = ax;
        e1y[seed] = ay;
        e2x[seed] = bx;
        e2y[seed] = by;
        occ[(GR_L / 2) * GR_L + GR_L / 2] = (uint8_t) (seed + 1);
        queue[0] = (uint8_t) seed;
    }
    qt = 1;
    GR_SYNC();
    // a node's four directions at once, a row of lanes each; the candidates still free in `rfree` (this lane's)
    uint32_t rfree = rvalid & ~bit_of(seed);
    while (qh < qt && !GR_CANCELLED(start)) {
        const uint32_t cur = queue[qh++];
        const int u0 = cu[cur], v0 = cv[cur];
        const double b1x = e1x[cur], b1y = e1y[cur], b2x = e2x[cur], b2y = e2y[cur];
        const double pcx = px[cur], pcy = py[cur];
        const int du = grp == 0u ? 1 : (grp == 1u ? -1 : 0), dv = grp == 2u ? 1 : (grp == 3u ? -1 : 0);
        const int u = u0 + du, v = v0 + dv;
        const bool cell_ok = !(u < -GR_L / 2 || u >= GR_L / 2 || v < -GR_L / 2 || v >= GR_L / 2) && occ[(v + GR_L / 2) * GR_L + (u + GR_L / 2)] == 0;
        const double sxp = du * b1x + dv * b2x, syp = du * b1y + dv * b2y;
        const double tx = pcx + sxp, ty = pcy + syp;
        // the vendored finder takes the nearest keypoint within minDistanceToAddKeypoint = 20 px of (neighbour + basis
        // vector) as the hole, else the line stays incomplete (circlesgrid.cpp:528,812-840,928-930: a synthetic
        // keypoint earns no existingVertexGain); capped at tol_frac of the step for small apparent patterns
        const double lim = fmin(tol_px * tol_px, tol_frac * tol_frac * (sxp * sxp + syp * syp));
        unsigned long long r = row_nearest(tx, ty, rfree, false, 0);
        // every row tests its own answer; the usual case — the accepted candidates are different ones — is written by the
        // four rows at once.  Two directions that want the same candidate go through the one-after-the-other form below.
        uint32_t jr = (uint32_t) (r & 0xFFu);
        bool acc = false;
        if (cell_ok && r != ~0ull) {
            const double ddx = px[jr] - tx, ddy = py[jr] - ty;
            acc = ddx * ddx + ddy * ddy <= lim;
        }
        const uint32_t a0 = (uint32_t) __builtin_amdgcn_readlane((int) acc, 0), a1 = (uint32_t) __builtin_amdgcn_readlane((int) acc, 16),
                       a2 = (uint32_t) __builtin_amdgcn_readlane((int) acc, 32), a3 = (uint32_t) __builtin_amdgcn_readlane((int) acc, 48);
        const uint32_t j0 = (uint32_t) __builtin_amdgcn_readlane((int) jr, 0), j1 = (uint32_t) __builtin_amdgcn_readlane((int) jr, 16),
                       j2 = (uint32_t) __builtin_amdgcn_readlane((int) jr, 32), j3 = (uint32_t) __builtin_amdgcn_readlane((int) jr, 48);
        const bool clash = (debug & 2) /* ECAL_FORCE=grid_serial_walk: always the one-after-the-other form */ || (a0 && a1 && j0 == j1) || (a0 && a2 && j0 == j2) || (a0 && a3 && j0 == j3) || (a1 && a2 && j1 == j2) ||
                           (a1 && a3 && j1 == j3) || (a2 && a3 && j2 == j3);
        uint32_t n_new = 0;
        if (!clash) {
            const uint32_t before = grp == 0u ? 0u : (grp == 1u ? a0 : (grp == 2u ? a0 + a1 : a0 + a1 + a2));
            n_new = a0 + a1 + a2 + a3;
            rfree &= ~((a0 ? bit_of(j0) : 0u) | (a1 ? bit_of(j1) : 0u) | (a2 ? bit_of(j2) : 0u) | (a3 ? bit_of(j3) : 0u));
            if (acc && gl == 0u) {
                const uint32_t j = jr;
                assigned[j] = 1;
                cu[j] = (int8_t) u;
                cv[j] = (int8_t) v;
                // the step actually taken refreshes the matching basis vector (perspective / distortion drift)
                const double mx = px[j] - pcx, my = py[j] - pcy;
                e1x[j] = du ? du * mx : b1x;
                e1y[j] = du ? du * my : b1y;
                e2x[j] = dv ? dv * mx : b2x;
                e2y[j] = dv ? dv * my : b2y;
                occ[(v + GR_L / 2) * GR_L + (u + GR_L / 2)] = (uint8_t) (j + 1);
                queue[qt + before] = (uint8_t) j;
            }
        } else {
        uint32_t taken[4];
#pragma unroll
        for (int d = 0; d < 4; d++) {   // in the order of the directions, as a walk that asks one question after the other
            taken[d] = 0xFFFFFFFFu;
            if (!__builtin_amdgcn_readlane((int) cell_ok, 16 * d)) continue;
            unsigned long long rd = lane_u64(r, 16 * d);
            if (rd == ~0ull) continue;
            uint32_t j = (uint32_t) (rd & 0xFFu);
            bool again = false;
#pragma unroll
            for (int e = 0; e < 4; e++) again = again || (e < d && taken[e] == j);
            if (again) {   // an earlier direction of this node took that candidate: this row asks again
                const unsigned long long r2 = row_nearest(tx, ty, rfree, false, 0);
                if (grp == (uint32_t) d) r = r2;
                rd = lane_u64(r2, 16 * d);
                if (rd == ~0ull) continue;
                j = (uint32_t) (rd & 0xFFu);
            }
            const double txd = lane_f64(tx, 16 * d), tyd = lane_f64(ty, 16 * d), limd = lane_f64(lim, 16 * d);
            const double ddx = px[j] - txd, ddy = py[j] - tyd;
            if (ddx * ddx + ddy * ddy > limd) continue;
            taken[d] = j;
            rfree &= ~bit_of(j);
            if (lane == 16u * (uint32_t) d) {
                assigned[j] = 1;
                cu[j] = (int8_t) u;
                cv[j] = (int8_t) v;
                const double mx = px[j] - pcx, my = py[j] - pcy;
                e1x[j] = du ? du * mx : b1x;
                e1y[j] = du ? du * my : b1y;
                e2x[j] = dv ? dv * mx : b2x;
                e2y[j] = dv ? dv * my : b2y;
                occ[(v + GR_L / 2) * GR_L + (u + GR_L / 2)] = (uint8_t) (j + 1);
                queue[qt + n_new] = (uint8_t) j;
            }
            n_new++;
        }
        }
        qt += n_new;
        GR_SYNC();
    }
    GR_SYNC();
    GR_MARK(1);   // first walk
    auto match_pattern = [&]() __attribute__((always_inline)) -> bool {
        if (qt < M) return false;
        // match the pattern: model point (x, y) = ((2j + i%2), i) has lattice coordinates U = (x+y)/2, V = (y-x)/2 in the
        // right-handed basis E1 = (1,1), E2 = (-1,1).  The walk's basis (two short independent steps around the seed) is SOME
        // right-handed basis of the same lattice: the diagonal steps in a frontal view — then the walk's coordinates are a
        // rotation of the model's —, but beyond ~55 degrees of tilt the foreshortened axis brings the second neighbour along it
        // closer than the diagonal ones and the basis comes out sheared.  So the model's coordinates go through every
        // T in SL2(Z) with entries in [-2, 2] (the four rotations first) and every visited node is tried as the image of model
        // point 0.  (This is what the vendored finder's second attempt on the homography-rectified points and its clustering
        // variant are for, cv_calib.cpp:34-84, circlesgrid.cpp:72-180: a complete pattern seen at a steep angle is still found.)
        // The model points' lattice coordinates are worked out once (mU, mV); the transforms go one after the other (the same
        // for the whole wave), a lane per anchor.  A transform whose image of the pattern is wider or taller than the box of
        // the visited cells cannot match whatever the anchor: most of the sheared ones end there, after a handful of
        // lane-uniform operations (the pattern's extent is that of its four corner points).
        if (lane == 0) {
            sh_box[0] = sh_box[2] = 127;
            sh_box[1] = sh_box[3] = -128;
        }
        GR_SYNC();
        for (uint32_t k = lane; k < qt; k += GR_T) {
            const uint32_t j = queue[k];
            atomicMin(&sh_box[0], (int) cu[j]);
            atomicMax(&sh_box[1], (int) cu[j]);
            atomicMin(&sh_box[2], (int) cv[j]);
            atomicMax(&sh_box[3], (int) cv[j]);
        }
        GR_SYNC();
        const int box_u = sh_box[1] - sh_box[0], box_v = sh_box[3] - sh_box[2];
        const uint32_t corner[3] = {cols - 1u, M - cols, M - 1u};
        uint32_t win_t = 0xFFFFFFFFu, win_anchor = 0;
        for (uint32_t t = 0; t < GR_NTF && win_t == 0xFFFFFFFFu && !GR_CANCELLED(start); t++) {   // (in order: the first match wins, rotations first)
            const int a = tf_sh[4 * t], b = tf_sh[4 * t + 1], c = tf_sh[4 * t + 2], d = tf_sh[4 * t + 3];
            int ulo = 0, uhi = 0, vlo = 0, vhi = 0;   // (model point 0 sits at the anchor)
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const int du = a * mU[corner[q]] + b * mV[corner[q]], dv = c * mU[corner[q]] + d * mV[corner[q]];
                ulo = min(ulo, du);
                uhi = max(uhi, du);
                vlo = min(vlo, dv);
                vhi = max(vhi, dv);
            }
            if (uhi - ulo > box_u || vhi - vlo > box_v) continue;
            for (uint32_t k0 = 0; k0 < qt && win_t == 0xFFFFFFFFu; k0 += GR_T) {
                const uint32_t k = k0 + lane;
                const uint32_t anchor_c = k < qt ? queue[k] : 0u;
                const int u0 = cu[anchor_c], v0 = cv[anchor_c];
                auto occupied = [&](uint32_t m) -> bool {
                    const int u = u0 + a * mU[m] + b * mV[m], v = v0 + c * mU[m] + d * mV[m];
                    return u >= -GR_L / 2 && u < GR_L / 2 && v >= -GR_L / 2 && v < GR_L / 2 &&
                           occ[(v + GR_L / 2) * GR_L + (u + GR_L / 2)] != 0;
                };
                // (a conjunction over the cells: the order of the tests is free — the far corners fail first for nearly every
                // wrong anchor, so the wave's slowest lane is done after three tests instead of a dozen)
                bool ok = k < qt && occupied(corner[2]) && occupied(corner[0]) && occupied(corner[1]);
                if (__ballot(ok) == 0ull) continue;
                for (uint32_t m = 0; m < M && ok; m++) ok = occupied(m);
                const unsigned long long hits = __ballot(ok);
                if (hits) {
                    win_t = t;
                    win_anchor = queue[k0 + (uint32_t) __ffsll((long long) hits) - 1u];
                }
            }
        }
        if (win_t == 0xFFFFFFFFu) return false;
        {
            const int a = tf_sh[4 * win_t], b = tf_sh[4 * win_t + 1], c = tf_sh[4 * win_t + 2], d = tf_sh[4 * win_t + 3];
            for (uint32_t m = lane; m < M; m += GR_T) {
                const int u = cu[win_anchor] + a * mU[m] + b * mV[m], v = cv[win_anchor] + c * mU[m] + d * mV[m];
                if constexpr (NW == 1) out[m] = (int32_t) occ[(v + GR_L / 2) * GR_L + (u + GR_L / 2)] - 1;   // (NW > 1: the winning start's wave writes it)
                sel[m] = (uint8_t) (occ[(v + GR_L / 2) * GR_L + (u + GR_L / 2)] - 1);
            }
        }
        return true;

    };
    got = match_pattern();
#ifdef ECAL_PHASE_PROF
    if (got) gr_outcome__ = start == 0 ? 0u : 2u;
#endif
    GR_MARK(2);   // first match
    // (the robust start gets its second attempt only when its first walk looks like the pattern's lattice: two thirds placed)
    if (!got && qt >= 4u && (start == 0 || (start == 1 && 3u * qt >= 2u * M))) {
        // Second attempt, as cv::findCirclesGrid's (cv_calib.cpp:34-84: the holes found so far give a homography,
        // CirclesGridFinder::rectifyGrid, and the search runs again on the rectified points).  Under steep perspective the
        // first walk's local steps drift and a step can land on the wrong neighbour (the foreshortened lattice has
        // neighbours closer than a step's prediction error); a homography's predictions are exact under perspective.
        // Only the seed's 3 x 3 neighbourhood of the first walk is kept (a step of that walk may have landed on the wrong
        // neighbour) and the walk is redone ring by ring with the homography's predictions, refitted after every ring.
        // (Keeping a first walk whose nodes all lie near one homography instead was measured: at 2 px it is 4 walks in 18 000 —
        // lens distortion —, at a third of a lattice step it changes nothing in the kernel's time.)
        if (lane == 0) {
            uint32_t keep = 0;
            for (uint32_t k = 0; k < qt; k++) {
                const uint32_t j = queue[k];
                if (cu[j] >= -1 && cu[j] <= 1 && cv[j] >= -1 && cv[j] <= 1) {
                    queue[keep++] = (uint8_t) j;
                } else {
                    assigned[j] = 0;
                    occ[(cv[j] + GR_L / 2) * GR_L + (cu[j] + GR_L / 2)] = 0;
                }
            }
            sh_qt = keep;
        }
        GR_SYNC();
        qt = sh_qt;
        for (int sweep = 0; sweep < 20 && !got && qt >= 4u && !GR_CANCELLED(start); sweep++) {
            GR_MARK(3);   // (restart, loop overhead)
#ifdef ECAL_PHASE_PROF
            gr_sweeps__++;
#endif
            fit_h();
            GR_MARK(4);   // homography fits
            if (!(hh[0] == hh[0])) break;
            const uint32_t qt_before = qt;
            // One ring: every open lattice cell next to a visited one is tried at the homography's prediction — all of them at
            // once, a lane per cell (the first walk's step vectors are dead: their arrays hold the cell list and the claims).
            // A cell takes the free candidate nearest to its prediction if that lies within the tolerance of a step from one of
            // the cell's visited neighbours (as in the first walk); a candidate wanted by two cells goes to the nearer one,
            // the other cell tries again in the next ring.  New nodes join the queue in row-major order of their cells.
            unsigned long long *const claim = reinterpret_cast<unsigned long long *>(e1x);   // [GR_MAXC]
            int8_t *const fl_u = reinterpret_cast<int8_t *>(e1y), *const fl_v = fl_u + 512;      // [<= 512] open cells of the ring
            if (lane == 0) {
                sh_box[0] = sh_box[2] = 127;
                sh_box[1] = sh_box[3] = -128;
            }
            for (uint32_t i = lane; i < n; i += GR_T) claim[i] = ~0ull;
            GR_SYNC();
            for (uint32_t k = lane; k < qt; k += GR_T) {
                const uint32_t j = queue[k];
                atomicMin(&sh_box[0], (int) cu[j]);
                atomicMax(&sh_box[1], (int) cu[j]);
                atomicMin(&sh_box[2], (int) cv[j]);
                atomicMax(&sh_box[3], (int) cv[j]);
            }
            GR_SYNC();
            const int vlo = max(sh_box[2] - 1, -GR_L / 2), vhi = min(sh_box[3] + 1, GR_L / 2 - 1);
            auto visited = [&](int u, int v) -> uint32_t {   // candidate index + 1 of the cell, 0: open or outside
                return (u < -GR_L / 2 || u >= GR_L / 2 || v < -GR_L / 2 || v >= GR_L / 2) ? 0u : occ[(v + GR_L / 2) * GR_L + (u + GR_L / 2)];
            };
            uint32_t nf = 0;   // the ring's open cells (two lattice rows of 32 per step)
            for (int v0 = vlo; v0 <= vhi; v0 += 2) {
                const int v = v0 + (int) (lane >> 5), u = (int) (lane & 31u) - GR_L / 2;
                const bool open = v <= vhi && visited(u, v) == 0u &&
                                  (visited(u + 1, v) | visited(u - 1, v) | visited(u, v + 1) | visited(u, v - 1)) != 0u;
                const unsigned long long m = __ballot(open);
                const uint32_t at = nf + (uint32_t) __popcll(m & ((1ull << lane) - 1ull));
                if (open && at < 512u) {
                    fl_u[at] = (int8_t) u;
                    fl_v[at] = (int8_t) v;
                }
                nf += (uint32_t) __popcll(m);
            }
            nf = nf < 512u ? nf : 512u;
            GR_SYNC();
            for (uint32_t k0 = 0; k0 < nf; k0 += GR_T) {
                const uint32_t k = k0 + lane;
                const bool mine = k < nf;
                const int u = mine ? fl_u[k] : 0, v = mine ? fl_v[k] : 0;
                const double wq = hh[6] * u + hh[7] * v + 1.0;
                const double tx = (hh[0] * u + hh[1] * v + hh[2]) / wq, ty = (hh[3] * u + hh[4] * v + hh[5]) / wq;
                double far2 = 0.0;   // the longest step to the cell from a visited neighbour
#pragma unroll
                for (int dir = 0; dir < 4; dir++) {
                    const int du = dir == 0 ? 1 : (dir == 1 ? -1 : 0), dv = dir == 2 ? 1 : (dir == 3 ? -1 : 0);
                    const uint32_t nb = mine ? visited(u + du, v + dv) : 0u;
                    if (nb) {
                        const double sx_ = tx - px[nb - 1u], sy_ = ty - py[nb - 1u];
                        far2 = fmax(far2, sx_ * sx_ + sy_ * sy_);
                    }
                }
                double best = INFINITY;
                uint32_t bj = 0;
                for (uint32_t i = 0; i < n; i++) {   // (the same i in every lane: broadcast reads)
                    if (assigned[i]) continue;
                    const double dx = px[i] - tx, dy = py[i] - ty, d2 = dx * dx + dy * dy;
                    if (d2 < best) {
                        best = d2;
                        bj = i;
                    }
                }
                const bool want = mine && wq > 1e-6 && best <= fmin(tol_px * tol_px, tol_frac * tol_frac * far2);
                // (distances order like their bit patterns; the low bits carry the cell: ties go to the earlier one)
                const unsigned long long key = (((unsigned long long) __double_as_longlong(best)) & ~0x3FFull) | (unsigned long long) k;
                if (want) atomicMin(&claim[bj], key);
                GR_SYNC();
                const bool won = want && claim[bj] == key;
                const unsigned long long wm = __ballot(won);
                if (won) {
                    const uint32_t at = qt + (uint32_t) __popcll(wm & ((1ull << lane) - 1ull));
                    assigned[bj] = 1;
                    cu[bj] = (int8_t) u;
                    cv[bj] = (int8_t) v;
                    occ[(v + GR_L / 2) * GR_L + (u + GR_L / 2)] = (uint8_t) (bj + 1);
                    queue[at] = (uint8_t) bj;
                }
                qt += (uint32_t) __popcll(wm);
                GR_SYNC();
            }
            GR_MARK(5);   // sweeps' searches
            if (qt == qt_before) break;
            if (qt >= M) got = match_pattern();
#ifdef ECAL_PHASE_PROF
            if (got) gr_outcome__ = start == 0 ? 1u : 3u;
#endif
            GR_MARK(6);   // matches after sweeps
        }
    }
    if ((debug & 1) && !got && lane == 0) out[8 + start] = -2000 - (int32_t) qt;   // (ECAL_TRACE=grid: nodes placed by a start that failed)
    if constexpr (NW > 1) {
        if (got && lane == 0) atomicMin(&best_start_sh, (uint32_t) start);
    }
    }   // (starts)
#undef GR_CANCELLED
#ifdef ECAL_PHASE_PROF
    gr_done__(qt);
#endif
    if constexpr (NW > 1) {
        __syncthreads();   // every wave is through its start
        const uint32_t win = best_start_sh;
        if (win == 255u || wv != win) return;   // (the lowest successful start's wave goes on: its `got` is true, its arrays hold the grid)
        for (uint32_t m = lane; m < M; m += GR_T) out[m] = (int32_t) sel[m];
    } else {
        if (!got) return;
    }
    // The holes, as the reference takes them: CirclesEventFrame.cpp:340-353 looks up the candidate nearest to every centre the
    // finder returns, and the finder's centres under clutter are the keypoints nearest the positions its basis predicts.  Here:
    // the homography model lattice -> image through the 36 matched candidates, refitted without the six worst (a hole a spurious
    // candidate took during the walk — its step predictions drift, a spurious point can be the nearer one — is among them),
    // and every model point takes the candidate nearest to its prediction; twice.  Without clutter nothing changes.
    for (int it = 0; it < 2; it++) {
        GR_SYNC();
        for (uint32_t m = lane; m < M; m += GR_T) {   // (the matched candidates are M different ones)
            const uint32_t j = sel[m];
            queue[m] = (uint8_t) j;
            cu[j] = mU[m];
            cv[j] = mV[m];
        }
        qt = M;
        GR_SYNC();
        fit_all();
        if (!(hh[0] == hh[0])) break;
        // residuals; the M - 6 best stay in the fit
        double *const res = e2x;   // [M] (the walk's step vectors are dead)
        for (uint32_t m = lane; m < M; m += GR_T) {
            const uint32_t j = sel[m];
            const double u = mU[m], v = mV[m], wq = hh[6] * u + hh[7] * v + 1.0;
            const double dx = (hh[0] * u + hh[1] * v + hh[2]) / wq - px[j], dy = (hh[3] * u + hh[4] * v + hh[5]) / wq - py[j];
            res[m] = dx * dx + dy * dy;
        }
        GR_SYNC();
        {   // a lane per model point counts the worse ones; the kept ones go to the queue in model order
            uint32_t keep = 0;
            for (uint32_t m0 = 0; m0 < M; m0 += GR_T) {
                const uint32_t m = m0 + lane;
                uint32_t worse = 0;   // model points with a larger residual (ties: larger index)
                if (m < M)
                    for (uint32_t q = 0; q < M; q++) worse += (res[q] > res[m] || (res[q] == res[m] && q > m)) ? 1u : 0u;
                const bool kept = m < M && (worse >= 6u || M <= 12u);
                const unsigned long long km = __ballot(kept);
                if (kept) queue[keep + (uint32_t) __popcll(km & ((1ull << lane) - 1ull))] = sel[m];
                keep += (uint32_t) __popcll(km);
            }
            qt = keep;
        }
        GR_SYNC();
        fit_all();
        if (!(hh[0] == hh[0])) break;
        // nearest candidates to the predictions; taken only when they are M distinct candidates
        uint8_t *const pick = assigned;   // [M] (dead too)
        bool distinct = true;
        for (uint32_t m = lane; m < M; m += GR_T) {   // a lane per model point looks at every candidate (broadcast reads; the same order: distance, then index)
            const double u = mU[m], v = mV[m], wq = hh[6] * u + hh[7] * v + 1.0;
            const double qx = (hh[0] * u + hh[1] * v + hh[2]) / wq, qy = (hh[3] * u + hh[4] * v + hh[5]) / wq;
            unsigned long long best = ~0ull;
            for (uint32_t i = 0; i < n; i++) {
                const double dx = px[i] - qx, dy = py[i] - qy;
                const unsigned long long k = pack_key(dx * dx + dy * dy, i);
                best = k < best ? k : best;
            }
            pick[m] = (uint8_t) (best & 0xFFu);
        }
        GR_SYNC();
        for (uint32_t m = lane; m < M; m += GR_T)
            for (uint32_t q = 0; q < m; q++)
                if (pick[q] == pick[m]) distinct = false;
        if (__ballot(!distinct) != 0ull) break;
        bool changed = false;
        for (uint32_t m = lane; m < M; m += GR_T) {
            changed = changed || sel[m] != pick[m];
            sel[m] = pick[m];
            out[m] = (int32_t) pick[m];
        }
        if (__ballot(changed) == 0ull) break;
    }
    if (lane == 0) found[s] = 1;
    // The keyframe search's next step — the rows' line fits of a window that holds a grid, the keyframe gate's input — here, by
    // the wave that has the grid (a lane per row, the same routine on the same values as the policy's own kernel): a kernel of
    // its own cost the search ~70 us of latency in every pass, while this launch lasts as long as its slowest FAILING window.
    if (dirs && rows <= GR_T) {
        GR_SYNC();
        if (lane < rows) {
            double dx, dy;
            row_direction(c, sel + lane * cols, cols, dx, dy);
            dirs[2 * ((size_t) s * rows + lane)] = dx;
            dirs[2 * ((size_t) s * rows + lane) + 1] = dy;
        }
    }
#undef GR_SYNC
}

}  // namespace ecal

using namespace ecal;

extern "C" int ecal_grid_order_dev(ecal_ctx *ctx, const uint32_t *d_win_info, const uint32_t *d_seg_off,
                                   const double *d_cand_xyr, uint32_t S, uint32_t rows, uint32_t cols,
                                   int32_t *d_order, uint32_t *d_found, void *stream) {
    return ecal_grid_order_dirs_dev(ctx, d_win_info, d_seg_off, d_cand_xyr, S, rows, cols, d_order, d_found, nullptr, stream);
}

// the same, and the found grids' row directions into d_dirs[S][rows][2] (may be NULL; ecal_adaptive.hip's passes)
int ecal_grid_order_dirs_dev(ecal_ctx *ctx, const uint32_t *d_win_info, const uint32_t *d_seg_off, const double *d_cand_xyr, uint32_t S,
                             uint32_t rows, uint32_t cols, int32_t *d_order, uint32_t *d_found, double *d_dirs, void *stream) {
    const ecal_range range__(ctx, "ecal_grid_order");
    if (!ctx) return ECAL_ERR_INVALID;
    if (S == 0) return ECAL_OK;
    if (!d_win_info || !d_seg_off || !d_cand_xyr || !d_order || !d_found || rows * cols < 4 || rows * cols > GR_MAXM) {
        ctx->last_error = "null pointer or unsupported pattern size";
        return ECAL_ERR_INVALID;
    }
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const double tol_px = ctx->sw.grid_tol_px;   // (ECAL_GRID_TOL_PX, a debug switch for tests of the tolerance's effect; default = the reference's 20 px)
    const int dbg = (ctx->sw.grid_debug ? 1 : 0) | (ctx->sw.grid_serial_walk ? 2 : 0);
    // few windows at work (the caller's word, ecal_ctx::grid_hint_windows — the keyframe search knows how many pieces are still
    // active —, else the launch's size): the latency form, a wave per start; ECAL_FORCE=grid_one_wave keeps the one-wave form
    const uint32_t at_work = ctx->grid_hint_windows ? ctx->grid_hint_windows : S;
    if (at_work <= GR_PARALLEL_MAX && !ctx->sw.grid_one_wave && !ctx->sw.grid_debug)
        hipLaunchKernelGGL(grid_order_kernel<4>, dim3(S), dim3(GR_T * 4), 0, (hipStream_t) stream, d_win_info, d_seg_off,
                           d_cand_xyr, rows, cols, 0.7, tol_px, d_order, d_found, dbg, d_dirs);
    else
        hipLaunchKernelGGL(grid_order_kernel<1>, dim3(S), dim3(GR_T), 0, (hipStream_t) stream, d_win_info, d_seg_off,
                           d_cand_xyr, rows, cols, 0.7, tol_px, d_order, d_found, dbg, d_dirs);
    ECAL_HIP_TRY(ctx, hipGetLastError());
    return ECAL_OK;
}

// Host-buffer form for ONE candidate list: what cv::findCirclesGrid(points, Size(cols, rows), centers, flags) is called with at
// CirclesEventFrame.cpp:332-336 (the shim host/cv_calib.hpp sits on this).  order[rows * cols] = candidate index per pattern point.
extern "C" int ecal_grid_order(ecal_ctx *ctx, const double *cand_xyr, uint32_t n, uint32_t rows, uint32_t cols, int32_t *order,
                               uint32_t *found) {
    if (!ctx || !order || !found || (n && !cand_xyr) || rows * cols < 4 || rows * cols > GR_MAXM) return ECAL_ERR_INVALID;
    const uint32_t M = rows * cols;
    *found = 0;
    for (uint32_t m = 0; m < M; m++) order[m] = -1;
    if (n < M || n > GR_MAXC) return ECAL_OK;      // (the kernel's own rule: fewer candidates than pattern points, or more than it handles)
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    ecal_devbuf *B = ctx->host_pipe;   // roles as in ecal_detect_pass: 13 win_info, 6 seg_off, 15 candidates
    int rc;
    if ((rc = ecal_ensure(ctx, B[13], 4 * sizeof(uint32_t))) || (rc = ecal_ensure(ctx, B[6], 2 * sizeof(uint32_t))) ||
        (rc = ecal_ensure(ctx, B[15], (size_t) n * 3 * sizeof(double))) || (rc = ecal_ensure(ctx, ctx->host_grid_order, M * sizeof(int32_t))) ||
        (rc = ecal_ensure(ctx, ctx->host_grid_found, sizeof(uint32_t))))
        return rc;
    const uint32_t info[4] = {n, 0, 0, 0}, off[2] = {0, 0};
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(B[13].ptr, info, sizeof(info), hipMemcpyHostToDevice, st));
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(B[6].ptr, off, sizeof(off), hipMemcpyHostToDevice, st));
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(B[15].ptr, cand_xyr, (size_t) n * 3 * sizeof(double), hipMemcpyHostToDevice, st));
    ECAL_HIP_TRY(ctx, hipStreamSynchronize(st));   // (the sources are the caller's pageable memory and this frame's locals)
    if ((rc = ecal_grid_order_dev(ctx, (const uint32_t *) B[13].ptr, (const uint32_t *) B[6].ptr, (const double *) B[15].ptr, 1, rows, cols,
                                  (int32_t *) ctx->host_grid_order.ptr, (uint32_t *) ctx->host_grid_found.ptr, st)))
        return rc;
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(order, ctx->host_grid_order.ptr, M * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(found, ctx->host_grid_found.ptr, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    ECAL_HIP_TRY(ctx, hipStreamSynchronize(st));
    return ECAL_OK;
}

#ifdef ECAL_PHASE_PROF
extern "C" int ecal_debug_grid_cycles(unsigned long long *out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(ecal::g_gr_cycles), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset == 4) {   // the outcomes instead
        if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(ecal::g_gr_out), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(ecal::g_gr_out), z, sizeof(z)) != hipSuccess) return -1;
        return 0;
    }
    if (reset >= 2) {   // the histogram instead
        unsigned long long hst[32];
        if (hipMemcpyFromSymbol(hst, HIP_SYMBOL(ecal::g_gr_hist), sizeof(hst)) != hipSuccess) return -1;
        for (int i = 0; i < 16; i++) out16[i] = hst[i + 10];   // 2^10 .. 2^23+ and the max in [14] -> out16[14]
        out16[14] = hst[24];
        unsigned long long z[32] = {0};
        if (reset == 3 && hipMemcpyToSymbol(HIP_SYMBOL(ecal::g_gr_hist), z, sizeof(z)) != hipSuccess) return -1;
        return 0;
    }
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(ecal::g_gr_cycles), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
