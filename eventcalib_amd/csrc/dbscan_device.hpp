// Device-side DBSCAN for one segment (one time-slice polarity), executed by one workgroup.
//
// Replaces the sequential reference pair
//   dbscan/include/dbscan.h:115-177,198-265  (Run / regionQuery / expandCluster)
//   dbscan/src/kdtree.cpp:106-179            (insertion-order kd-tree, range query)
// with a data-parallel formulation that yields the same labels bit for bit:
//   B. kd-cell bounds: a level-synchronous replay of kd_insert gives every point the deepest
//      ancestor per dimension whose right (lo) / left (hi) subtree contains it;
//   C. torus cell hash (cell edge a hair above eps) + counting sort of the points by bucket;
//   D. neighbour count  = #{ j != i : in-ball(i,j) and not pruned(i,j) }   -> core flags;
//   E. min-seed directed reachability over core points (push/pull sweeps + pointer jumping);
//   F. rank of the seeds in pid order = the reference's cluster numbering.
// Arithmetic of the ball predicate is the reference's: (xj-xi)^2 + (yj-yi)^2 <= eps*eps with
// separate mul/add (kdtree.cpp:155-159) — this translation unit must be built -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace ecal {

constexpr uint32_t NONE32 = 0xFFFFFFFFu;

template <typename Idx>
struct IdxBits;
template <>
struct IdxBits<uint16_t> {  // LDS tiers: node ids < 8192
    static constexpr uint32_t NONE = 0xFFFFu;
    static constexpr uint32_t PLACED = 0x8000u, SIDE = 0x4000u, DIR = 0x2000u, MASK = 0x1FFFu;
};
template <>
struct IdxBits<uint32_t> {  // global-scratch tier: node ids < 2^29
    static constexpr uint32_t NONE = 0xFFFFFFFFu;
    static constexpr uint32_t PLACED = 0x80000000u, SIDE = 0x40000000u, DIR = 0x20000000u, MASK = 0x1FFFFFFFu;
};

// Segment workspace.  Pointers are LDS (tiers) or global scratch (big tier).
template <typename Idx>
struct DbWork {
    const double *c;  // [2n] coordinates, interleaved (x0,y0,x1,y1,...) exactly as the caller's xy
    uint32_t *slot;   // [2n+2]  phase B: child table; later: label[n] | bucket cursor[nb+1]
    Idx *anc;         // [4n]   lo_x, lo_y, hi_x, hi_y ancestor ids per point
    Idx *cur;         // [n]    phase B cursor; later: points sorted by bucket; finally seed ranks
    uint32_t *red;    // [>= T/64 + 1] scratch for block scans (always LDS)
};

template <bool GLOBAL>
__device__ __forceinline__ uint32_t ld_shared_word(const uint32_t *p) {
    if constexpr (GLOBAL) {
        // written by other waves' atomics: read at L2, never from this CU's L1
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        return *p;
    }
}

template <bool GLOBAL, typename Idx>
__device__ __forceinline__ uint32_t ld_idx(const Idx *p) {
    if constexpr (GLOBAL) {
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        return *p;
    }
}

struct Grid {
    double inv_cell;
    uint32_t bx_mask, by_mask, bx_log;  // torus BX x BY buckets, both powers of two >= 4
    __device__ __forceinline__ uint32_t bucket(int cx, int cy) const {
        return ((uint32_t) cx & bx_mask) | (((uint32_t) cy & by_mask) << bx_log);
    }
};

// exclusive block scan of one value per thread; returns the exclusive prefix, *total = sum.
template <int T>
__device__ __forceinline__ uint32_t block_exscan(uint32_t v, uint32_t *red, uint32_t *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) red[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < T / 64; w++) {
        uint32_t s = red[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// pruned(i -> j): the reference's range query started at point i never reaches node j although
// j is inside the ball, because an ancestor of j prunes j's side (kdtree.cpp:166-172).
// d selects the dimension whose |delta| equals eps exactly (the only case that can prune).
template <typename Idx>
__device__ __forceinline__ bool pruned_dim(const double *c, const Idx *anc, uint32_t j, uint32_t d, double q,
                                           double eps) {
    const uint32_t lo = anc[4 * j + d];
    if (lo != IdxBits<Idx>::NONE) {
        const double v = c[2 * lo + d];
        if (q <= v && !(fabs(q - v) < eps)) return true;
    }
    const uint32_t hi = anc[4 * j + 2 + d];
    if (hi != IdxBits<Idx>::NONE) {
        const double v = c[2 * hi + d];
        if (q > v && !(fabs(q - v) < eps)) return true;
    }
    return false;
}

// Visit every candidate j of point i (all points of the 3x3 cell block, or all points when the
// grid is unusable) and call f(j, xj, yj).
template <bool GLOBAL, typename Idx, typename F>
__device__ __forceinline__ void for_candidates(const double *c, const Idx *sorted, uint32_t n, bool use_grid,
                                               const Grid &g, const uint32_t *bend, double xi, double yi, F &&f) {
    const double2 *c2 = reinterpret_cast<const double2 *>(c);
    if (!use_grid) {
        for (uint32_t j = 0; j < n; j++) {
            const double2 p = c2[j];
            f(j, p.x, p.y);
        }
        return;
    }
    const int cx = (int) floor(xi * g.inv_cell), cy = (int) floor(yi * g.inv_cell);
#pragma unroll 1
    for (int oy = -1; oy <= 1; oy++) {
#pragma unroll 1
        for (int ox = -1; ox <= 1; ox++) {
            const uint32_t b = g.bucket(cx + ox, cy + oy);
            uint32_t k = b ? ld_shared_word<GLOBAL>(&bend[b - 1]) : 0u;
            const uint32_t e = ld_shared_word<GLOBAL>(&bend[b]);
            for (; k < e; k++) {
                const uint32_t j = sorted[k];
                const double2 p = c2[j];
                f(j, p.x, p.y);
            }
        }
    }
}

// The whole pipeline for one segment.  Preconditions: wk.c holds the n points (pid order),
// all threads of the block call this with identical arguments, n >= 1, minpts >= 1.
// out_labels: global, [n].  Returns the cluster count (valid on every thread).
template <int T, bool GLOBAL, typename Idx>
__device__ __forceinline__ uint32_t dbscan_segment(const DbWork<Idx> wk, uint32_t n, double eps, uint32_t minpts,
                                                   uint32_t nb_log, int32_t *out_labels) {
    using B = IdxBits<Idx>;
    const uint32_t tid = threadIdx.x;
    const double *const C = wk.c;
    const double2 *const C2 = reinterpret_cast<const double2 *>(wk.c);
    uint32_t *const slot = wk.slot;
    Idx *const anc = wk.anc;
    Idx *const cur = wk.cur;
    uint32_t *const red = wk.red;

    // ---------------- B: kd-cell bounds by level-synchronous insertion replay ----------------
    for (uint32_t i = tid; i < n; i += T) {
        cur[i] = (Idx) (i == 0 ? B::PLACED : 0u);  // compare against root (node 0), dir 0
        anc[4 * i + 0] = (Idx) B::NONE;
        anc[4 * i + 1] = (Idx) B::NONE;
        anc[4 * i + 2] = (Idx) B::NONE;
        anc[4 * i + 3] = (Idx) B::NONE;
        slot[2 * i] = NONE32;
        slot[2 * i + 1] = NONE32;
    }
    __syncthreads();
    for (;;) {
        for (uint32_t i = tid; i < n; i += T) {
            uint32_t c = cur[i];
            if (c & B::PLACED) continue;
            const uint32_t a = c & B::MASK;
            const uint32_t d = (c & B::DIR) ? 1u : 0u;
            const uint32_t side = (C[2 * i + d] < C[2 * a + d]) ? 0u : 1u;  // kdtree.cpp:128 — ties go right
            atomicMin(&slot[2 * a + side], i);
            cur[i] = (Idx) (c | (side ? B::SIDE : 0u));
        }
        __syncthreads();
        int active = 0;
        for (uint32_t i = tid; i < n; i += T) {
            uint32_t c = cur[i];
            if (c & B::PLACED) continue;
            const uint32_t a = c & B::MASK;
            const uint32_t d = (c & B::DIR) ? 1u : 0u;
            const uint32_t side = (c & B::SIDE) ? 1u : 0u;
            const uint32_t child = ld_shared_word<GLOBAL>(&slot[2 * a + side]);
            anc[4 * i + (side ? 0 : 2) + d] = (Idx) a;  // deepest ancestor wins
            if (child == i) {
                cur[i] = (Idx) B::PLACED;
            } else {
                cur[i] = (Idx) (child | (d ? 0u : B::DIR));
                active = 1;
            }
        }
        if (!__syncthreads_or(active)) break;
    }

    // ---------------- C: torus cell hash + counting sort ----------------
    uint32_t *label = slot;         // [n]
    uint32_t *bend = slot + n + 1;  // [nb]   after the scatter: end offset of each bucket
    const uint32_t nb = 1u << nb_log;
    Grid g;
    g.bx_log = (nb_log + 1) >> 1;
    g.bx_mask = (1u << g.bx_log) - 1u;
    g.by_mask = (1u << (nb_log - g.bx_log)) - 1u;
    // cell edge = eps * (1 + 2^-20): two points within eps (as the fp predicate sees it) are
    // always in adjacent cells as long as |coord| / eps < 2^30 (checked below).
    const double cell = eps * 1.00000095367431640625;
    g.inv_cell = 1.0 / cell;
    int bad = 0;
    if (!(eps > 0.0) || !(g.inv_cell < 1.0e300)) bad = 1;
    for (uint32_t i = tid; i < n; i += T) {
        const double2 p = C2[i];
        const double fx = floor(p.x * g.inv_cell), fy = floor(p.y * g.inv_cell);
        if (!(fabs(fx) < 1073741824.0) || !(fabs(fy) < 1073741824.0)) bad = 1;  // also catches NaN/inf
    }
    for (uint32_t b = tid; b < nb; b += T) bend[b] = 0;
    const bool use_grid = !__syncthreads_or(bad);
    if (use_grid) {
        for (uint32_t i = tid; i < n; i += T) {
            const double2 p = C2[i];
            const int cx = (int) floor(p.x * g.inv_cell), cy = (int) floor(p.y * g.inv_cell);
            atomicAdd(&bend[g.bucket(cx, cy)], 1u);
        }
        __syncthreads();
        // exclusive scan of the bucket counts (blocked: each thread owns nb/T consecutive buckets)
        {
            const uint32_t per = (nb + T - 1) / T;
            const uint32_t b0 = tid * per;
            uint32_t sum = 0;
            for (uint32_t b = b0; b < b0 + per && b < nb; b++) sum += ld_shared_word<GLOBAL>(&bend[b]);
            uint32_t total;
            uint32_t run = block_exscan<T>(sum, red, &total);
            for (uint32_t b = b0; b < b0 + per && b < nb; b++) {
                const uint32_t c = ld_shared_word<GLOBAL>(&bend[b]);
                bend[b] = run;
                run += c;
            }
        }
        __syncthreads();
        for (uint32_t i = tid; i < n; i += T) {
            const double2 p = C2[i];
            const int cx = (int) floor(p.x * g.inv_cell), cy = (int) floor(p.y * g.inv_cell);
            const uint32_t pos = atomicAdd(&bend[g.bucket(cx, cy)], 1u);
            cur[pos] = (Idx) i;
        }
        __syncthreads();
    }

    // ---------------- D: neighbour counts -> core flags ----------------
    const double eps2 = eps * eps;  // SQ(range), kdtree.cpp:159
    const double aeps = fabs(eps);
    for (uint32_t i = tid; i < n; i += T) {
        const double xi = C2[i].x, yi = C2[i].y;
        uint32_t cnt = 0;
        for_candidates<GLOBAL>(C, cur, n, use_grid, g, bend, xi, yi, [&](uint32_t j, double xj, double yj) {
            const double dx = xj - xi, dy = yj - yi;  // node.pos - query.pos
            const double d2 = dx * dx + dy * dy;
            if (d2 <= eps2 && j != i) {
                bool pr = false;
                if (fabs(dx) == aeps) pr = pruned_dim(C, anc, j, 0u, xi, eps);
                if (!pr && fabs(dy) == aeps) pr = pruned_dim(C, anc, j, 1u, yi, eps);
                cnt += pr ? 0u : 1u;
            }
        });
        label[i] = (cnt >= minpts) ? i : NONE32;
    }
    __syncthreads();

    // ---------------- E: min-seed directed reachability over core points ----------------
    for (;;) {
        int changed = 0;
        for (uint32_t i = tid; i < n; i += T) {
            const uint32_t L0 = ld_shared_word<GLOBAL>(&label[i]);
            if (L0 == NONE32) continue;
            uint32_t L = L0;
            const double xi = C2[i].x, yi = C2[i].y;
            for_candidates<GLOBAL>(C, cur, n, use_grid, g, bend, xi, yi, [&](uint32_t j, double xj, double yj) {
                const double dx = xj - xi, dy = yj - yi;
                const double d2 = dx * dx + dy * dy;
                if (d2 <= eps2 && j != i) {
                    const uint32_t Lj = ld_shared_word<GLOBAL>(&label[j]);
                    if (Lj == NONE32) return;
                    bool fwd = true, bwd = true;  // edge i->j, edge j->i
                    if (fabs(dx) == aeps) {
                        if (pruned_dim(C, anc, j, 0u, xi, eps)) fwd = false;
                        if (pruned_dim(C, anc, i, 0u, xj, eps)) bwd = false;
                    }
                    if (fabs(dy) == aeps) {
                        if (fwd && pruned_dim(C, anc, j, 1u, yi, eps)) fwd = false;
                        if (bwd && pruned_dim(C, anc, i, 1u, yj, eps)) bwd = false;
                    }
                    if (bwd && Lj < L) L = Lj;
                    if (fwd && L < Lj) {
                        atomicMin(&label[j], L);
                        changed = 1;
                    }
                }
            });
            if (L < L0) {
                atomicMin(&label[i], L);
                changed = 1;
            }
        }
        __syncthreads();
        // pointer jumping: label[v] always names a core point that reaches v, so does label[label[v]]
        for (uint32_t i = tid; i < n; i += T) {
            uint32_t L = ld_shared_word<GLOBAL>(&label[i]);
            if (L == NONE32) continue;
            uint32_t L0 = L;
            for (;;) {
                const uint32_t up = ld_shared_word<GLOBAL>(&label[L]);
                if (up >= L) break;
                L = up;
            }
            if (L < L0) atomicMin(&label[i], L);
        }
        if (!__syncthreads_or(changed)) break;
    }

    // ---------------- F: seeds ranked in pid order = reference cluster ids ----------------
    uint32_t total;
    {
        const uint32_t per = (n + T - 1) / T;
        const uint32_t i0 = tid * per;
        uint32_t mine = 0;
        for (uint32_t i = i0; i < i0 + per && i < n; i++) mine += (ld_shared_word<GLOBAL>(&label[i]) == i) ? 1u : 0u;
        uint32_t run = block_exscan<T>(mine, red, &total);
        for (uint32_t i = i0; i < i0 + per && i < n; i++) {
            if (ld_shared_word<GLOBAL>(&label[i]) == i) cur[i] = (Idx) (run++);
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < n; i += T) {
        const uint32_t L = ld_shared_word<GLOBAL>(&label[i]);
        out_labels[i] = (L == NONE32) ? -1 : (int32_t) ld_idx<GLOBAL>(&cur[L]);
    }
    __syncthreads();
    return total;
}

}  // namespace ecal
