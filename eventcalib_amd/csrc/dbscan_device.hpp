// Device-side DBSCAN for one segment (one time-slice polarity), executed by one workgroup.
//
// Replaces the sequential reference pair
//   dbscan/include/dbscan.h:115-177,198-265  (Run / regionQuery / expandCluster)
//   dbscan/src/kdtree.cpp:106-179            (insertion-order kd-tree, range query)
// with a data-parallel formulation that yields the same labels bit for bit (design/03_dbscan.md):
//   B. kd-cell bounds: a level-synchronous replay of kd_insert gives every point the deepest
//      ancestor per dimension whose right (lo) / left (hi) subtree contains it;
//   C. torus cell hash (cell edge >= eps); points counting-sorted by bucket, their coordinates
//      stored in that order so a candidate costs one LDS read;
//   D. neighbour count  = #{ j != i : in-ball(i,j) and not pruned(i,j) }, stopped at minpts;
//   E. min-seed directed reachability over core points: one sweep of lock-free union-find over
//      the two-way edges, then a fix-point over the (rare) one-way edges the pruning creates;
//   F. rank of the seeds in pid order = the reference's cluster numbering.
// Two geometry policies share the pipeline: GeoF64 evaluates the predicates in the reference's
// own double arithmetic; GeoI16 is taken (per segment, decided on the device) when every
// coordinate is an integer of magnitude <= 16383 — event pixels — and evaluates the SAME
// predicates in exact 32-bit integer arithmetic (double arithmetic on such values is exact, so
// every comparison has the identical outcome; design/03_dbscan.md).
// Arithmetic of the ball predicate is the reference's: (xj-xi)^2 + (yj-yi)^2 <= eps*eps with
// separate mul/add (kdtree.cpp:155-159) — this translation unit must be built -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace ecal {

#ifdef ECAL_PHASE_PROF
// debug build only: shader-clock cycles spent per phase, summed over workgroups (thread 0)
__device__ unsigned long long g_phase_cycles[16];
#define ECAL_PHASE_MARK(idx)                                                            \
    do {                                                                                \
        if (threadIdx.x == 0) {                                                         \
            const unsigned long long now__ = __builtin_readcyclecounter();              \
            atomicAdd(&g_phase_cycles[idx], now__ - phase_t__);                         \
            phase_t__ = now__;                                                          \
        }                                                                               \
    } while (0)
#define ECAL_PHASE_COUNT(idx, v)                                                        \
    do {                                                                                \
        if (threadIdx.x == 0) atomicAdd(&g_phase_cycles[idx], (unsigned long long) (v)); \
    } while (0)
#else
#define ECAL_PHASE_MARK(idx) do { } while (0)
#define ECAL_PHASE_COUNT(idx, v) do { } while (0)
#endif

constexpr uint32_t NONE32 = 0xFFFFFFFFu;
constexpr uint32_t EDGE_CAP = 256;  // one-way edges kept per segment before falling back to sweeps
constexpr uint32_t KTOP = 64;       // insertions replayed by wave 0 alone (the top of the kd-tree)

template <typename Idx>
struct IdxBits;
template <>
struct IdxBits<uint16_t> {  // LDS tiers: node ids < 8192
    static constexpr uint32_t NONE = 0xFFFFu;
    static constexpr uint32_t PLACED = 0x8000u, SIDE = 0x4000u, DIR = 0x2000u, MASK = 0x1FFFu;
};
template <>
struct IdxBits<uint32_t> {  // global-scratch tier / register state: node ids < 2^29
    static constexpr uint32_t NONE = 0xFFFFFFFFu;
    static constexpr uint32_t PLACED = 0x80000000u, SIDE = 0x40000000u, DIR = 0x20000000u, MASK = 0x1FFFFFFFu;
};

// ---------------- geometry policies ----------------
struct GeoF64 {
    using Store = double2;  // one point in the workspace
    using S = double;       // coordinate / delta scalar
    double eps, eps2, aeps, inv_cell;
    __device__ __forceinline__ void init(double e) {
        eps = e;
        eps2 = e * e;  // SQ(range), kdtree.cpp:159
        aeps = fabs(e);
        // cell edge = eps * (1 + 2^-20): two points within eps (as the fp predicate sees it) are
        // always in adjacent cells as long as |coord| / eps < 2^30 (cell_ok).
        inv_cell = 1.0 / (e * 1.00000095367431640625);
    }
    static __device__ __forceinline__ Store pack(double2 p) { return p; }
    static __device__ __forceinline__ double2 unpack(Store s) { return s; }
    static __device__ __forceinline__ S sx(Store s) { return s.x; }
    static __device__ __forceinline__ S sy(Store s) { return s.y; }
    __device__ __forceinline__ bool in_ball(S dx, S dy) const { return dx * dx + dy * dy <= eps2; }
    __device__ __forceinline__ bool pos_eps(S d) const { return d == aeps; }
    __device__ __forceinline__ bool neg_eps(S d) const { return d == -aeps; }
    __device__ __forceinline__ int cell(S v) const { return (int) floor(v * inv_cell); }
    __device__ __forceinline__ bool cell_ok(S v) const { return fabs(floor(v * inv_cell)) < 1073741824.0; }
    __device__ __forceinline__ bool usable() const { return inv_cell < 1.0e300; }
};

struct GeoI16 {
    using Store = uint32_t;  // x | y << 16, two's complement int16 each
    using S = int;
    int e2i, epsi, cshift;
    __device__ __forceinline__ void init(double e) {
        const double e2 = e * e;
        e2i = (e2 < 2147483647.0) ? (int) floor(e2) : 2147483647;          // d2 <= eps^2  <=>  d2 <= floor(eps^2)
        epsi = (e == floor(e) && e <= 32767.0) ? (int) e : 0x40000000;     // |delta| == eps needs an integral eps
        cshift = 0;
        while (cshift < 16 && (double) (1 << cshift) < e) cshift++;       // cell edge 2^cshift >= eps
    }
    static __device__ __forceinline__ bool fits(double2 p) {
        return p.x == floor(p.x) && p.y == floor(p.y) && fabs(p.x) <= 16383.0 && fabs(p.y) <= 16383.0;
    }
    static __device__ __forceinline__ Store pack(double2 p) {
        return ((uint32_t) (int) p.x & 0xFFFFu) | ((uint32_t) (int) p.y << 16);
    }
    static __device__ __forceinline__ S sx(Store s) { return (int) (short) (s & 0xFFFFu); }
    static __device__ __forceinline__ S sy(Store s) { return ((int) s) >> 16; }
    static __device__ __forceinline__ double2 unpack(Store s) { return make_double2((double) sx(s), (double) sy(s)); }
    __device__ __forceinline__ bool in_ball(S dx, S dy) const { return __mul24(dx, dx) + __mul24(dy, dy) <= e2i; }
    __device__ __forceinline__ bool pos_eps(S d) const { return d == epsi; }
    __device__ __forceinline__ bool neg_eps(S d) const { return d == -epsi; }
    __device__ __forceinline__ int cell(S v) const { return v >> cshift; }
    __device__ __forceinline__ bool cell_ok(S) const { return true; }
    __device__ __forceinline__ bool usable() const { return true; }
};

// Segment workspace.  Pointers are LDS (tiers) or global scratch (big tier).
template <typename Idx, typename G>
struct DbWork {
    const typename G::Store *p;  // [n] points in pid order, read during B/C
    typename G::Store *cs;       // [n] points in bucket order, filled in C (LDS tiers: aliases p)
    uint32_t *slot;              // [2n+4] B: child table; later: parent/label[n] | bucket starts[nb+2]
    Idx *anc;                    // [4n] lo_x, lo_y, hi_x, hi_y ancestor pids; E.3: component labels (u32[n])
    Idx *pid_s;                  // [n] B (memory variant): cursor; C..E: bucket position -> pid; F: seed ranks
    Idx *inv;                    // [n] pid -> bucket position
    uint8_t *sflags;             // [n] by bucket position: prune-filter bits (point_flags) | 16 = core
    uint8_t *pflags;             // [n] global tier only: the same bits by pid until the scatter
    uint32_t *red;               // [48] block-scan scratch, edge counter, block_any flags (always LDS)
    uint32_t *edges;             // [2*EDGE_CAP] one-way edges (src pid, dst pid) (always LDS)
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

struct Grid {
    uint32_t bx_mask, by_mask, bx_log;  // torus BX x BY buckets, both powers of two >= 4
    __device__ __forceinline__ uint32_t bucket(int cx, int cy) const {
        return ((uint32_t) cx & bx_mask) | (((uint32_t) cy & by_mask) << bx_log);
    }
};

// Workgroup-wide "does any thread have pred set?" with ONE barrier per call (HIP's
// __syncthreads_or costs three).  flags = 3 LDS words, zero before the first call; `round` must
// count up by one per call: round r raises flags[r%3], while flags[(r+1)%3] — last read two
// barriers ago — is cleared for the next round.
__device__ __forceinline__ bool block_any(bool pred, uint32_t *flags, uint32_t &round) {
    const uint32_t r = round % 3u, nx = (round + 1u) % 3u;
    if (threadIdx.x == 0) flags[nx] = 0;
    if (__any(pred) && (threadIdx.x & 63) == 0) flags[r] = 1;
    __syncthreads();
    round++;
    return flags[r] != 0;
}

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

// pruned(i -> j) in dimension d: the reference's range query started at point i (coordinate q in
// that dimension) never reaches node j although j is inside the ball, because an ancestor of j
// prunes j's side (kdtree.cpp:166-172).  Only possible when |delta_d| == eps exactly (DESIGN.md
// §3.D).  Always evaluated in the reference's double arithmetic.
// cs/inv: coordinates in bucket order + pid -> position, or (inv == nullptr) coordinates in pid order.
template <typename G, typename Idx>
__device__ __forceinline__ bool pruned_dim(const typename G::Store *cs, const Idx *inv, const Idx *anc, uint32_t pj,
                                           uint32_t d, double q, double eps) {
    const uint32_t lo = anc[4 * pj + d];
    if (lo != IdxBits<Idx>::NONE) {
        const double2 a = G::unpack(cs[inv ? (uint32_t) inv[lo] : lo]);
        const double v = d ? a.y : a.x;
        if (q <= v && !(fabs(q - v) < eps)) return true;
    }
    const uint32_t hi = anc[4 * pj + 2 + d];
    if (hi != IdxBits<Idx>::NONE) {
        const double2 a = G::unpack(cs[inv ? (uint32_t) inv[hi] : hi]);
        const double v = d ? a.y : a.x;
        if (q > v && !(fabs(q - v) < eps)) return true;
    }
    return false;
}

// Filter bits of point j: bit0/bit1 = its lo_x/lo_y ancestor lies within eps*2^-30 below j's own
// coordinate, bit2/bit3 = its hi_x/hi_y ancestor within eps*2^-30 above.  pruned_dim(.., j, d, ..)
// can only fire through an ancestor whose bit is set: the pruning ancestor's coordinate and j's
// both round to a difference of exactly eps from the query, so they are within one ulp(eps) of
// each other (design/03_dbscan.md).  For pixel data the bit simply says "an ancestor shares my x (y)".
template <typename G, typename Idx>
__device__ __forceinline__ uint32_t point_flags(const typename G::Store *p, typename G::Store self, uint32_t lox,
                                                uint32_t loy, uint32_t hix, uint32_t hiy, double tol) {
    const double2 pi = G::unpack(self);
    uint32_t f = 0;
    if (lox != IdxBits<Idx>::NONE && (pi.x - G::unpack(p[lox]).x) < tol) f |= 1u;
    if (loy != IdxBits<Idx>::NONE && (pi.y - G::unpack(p[loy]).y) < tol) f |= 2u;
    if (hix != IdxBits<Idx>::NONE && (G::unpack(p[hix]).x - pi.x) < tol) f |= 4u;
    if (hiy != IdxBits<Idx>::NONE && (G::unpack(p[hiy]).y - pi.y) < tol) f |= 8u;
    return f;
}

// may the reference's query from i miss j (delta = j - i = (dx,dy), fj = j's filter bits)?
template <typename G>
__device__ __forceinline__ bool maybe_pruned(const G &geo, typename G::S dx, typename G::S dy, uint32_t fj) {
    return (geo.pos_eps(dx) && (fj & 1u)) || (geo.neg_eps(dx) && (fj & 4u)) || (geo.pos_eps(dy) && (fj & 2u)) ||
           (geo.neg_eps(dy) && (fj & 8u));
}

// Exact classification of the pair (query i at q, candidate j at c), both inside the ball: does
// the reference's range query from i return j (fwd) and from j return i (bwd)?
template <typename G, typename Idx>
__device__ __forceinline__ void edge_dirs(const typename G::Store *cs, const Idx *inv, const Idx *anc, uint32_t pi,
                                          uint32_t pj, double2 q, double2 c, double eps, bool &fwd, bool &bwd) {
    const double dx = c.x - q.x, dy = c.y - q.y, aeps = fabs(eps);
    fwd = true;
    bwd = true;
    if (fabs(dx) == aeps) {
        if (pruned_dim<G>(cs, inv, anc, pj, 0u, q.x, eps)) fwd = false;
        if (pruned_dim<G>(cs, inv, anc, pi, 0u, c.x, eps)) bwd = false;
    }
    if (fabs(dy) == aeps) {
        if (fwd && pruned_dim<G>(cs, inv, anc, pj, 1u, q.y, eps)) fwd = false;
        if (bwd && pruned_dim<G>(cs, inv, anc, pi, 1u, c.y, eps)) bwd = false;
    }
}

// Visit the candidates of a query at (xq, yq): every point of the 3x3 cell block (three
// bucket-contiguous row ranges, split in two where the torus wraps), or every point when the grid
// is unusable.  f(k, store_k, flags_k) returns false to stop the whole visit.
template <bool GLOBAL, typename G, typename F>
__device__ __forceinline__ void for_candidates(const G &geo, const typename G::Store *cs, const uint8_t *sflags,
                                               uint32_t n, bool use_grid, const Grid &g, const uint32_t *bstart,
                                               typename G::S xq, typename G::S yq, F &&f) {
    if (!use_grid) {
        for (uint32_t k = 0; k < n; k++)
            if (!f(k, cs[k], (uint32_t) sflags[k])) return;
        return;
    }
    const int cx = geo.cell(xq), cy = geo.cell(yq);
    const uint32_t x0 = (uint32_t) (cx - 1) & g.bx_mask;
    const uint32_t x1 = min(x0 + 2u, g.bx_mask);
    const bool wrap = x0 + 2u > g.bx_mask;
    const uint32_t wx = x0 + 2u - g.bx_mask - 1u;  // last wrapped column (meaningful when wrap)
#pragma unroll 1
    for (int oy = -1; oy <= 1; oy++) {
        const uint32_t rb = ((uint32_t) (cy + oy) & g.by_mask) << g.bx_log;
        uint32_t k = ld_shared_word<GLOBAL>(&bstart[rb + x0]);
        uint32_t e = ld_shared_word<GLOBAL>(&bstart[rb + x1 + 1]);
        for (; k < e; k++)
            if (!f(k, cs[k], (uint32_t) sflags[k])) return;
        if (wrap) {
            k = ld_shared_word<GLOBAL>(&bstart[rb]);
            e = ld_shared_word<GLOBAL>(&bstart[rb + wx + 1]);
            for (; k < e; k++)
                if (!f(k, cs[k], (uint32_t) sflags[k])) return;
        }
    }
}

// lock-free union-find over pids; roots have parent[r] == r; the smaller pid always wins the root,
// so every parent pointer leads to a smaller pid and the structure stays acyclic under races.
template <bool GLOBAL>
__device__ __forceinline__ uint32_t uf_find(uint32_t *parent, uint32_t v) {
    uint32_t p = ld_shared_word<GLOBAL>(&parent[v]);
    while (p != v) {
        const uint32_t gp = ld_shared_word<GLOBAL>(&parent[p]);
        if (gp != p) parent[v] = gp;  // path halving: any ancestor is a valid parent
        v = p;
        p = gp;
    }
    return v;
}

// Root of v without touching the forest.  The flatten pass (E.2) must use this one: with path halving, a thread that
// read parent[v] before another thread stored v's root there would overwrite that root with a mere ancestor, and v
// would come out of the pass unflattened (seen as a wrong label in ~1 of 10^5 segments).  Without the halving stores
// every store of the pass is a true root, and a walk that meets one simply gets there sooner.
template <bool GLOBAL>
__device__ __forceinline__ uint32_t uf_root(const uint32_t *parent, uint32_t v) {
    uint32_t p = ld_shared_word<GLOBAL>(&parent[v]);
    while (p != v) {
        v = p;
        p = ld_shared_word<GLOBAL>(&parent[p]);
    }
    return v;
}

template <bool GLOBAL>
__device__ __forceinline__ void uf_union(uint32_t *parent, uint32_t a, uint32_t b) {
    for (;;) {
        a = uf_find<GLOBAL>(parent, a);
        b = uf_find<GLOBAL>(parent, b);
        if (a == b) return;
        if (a < b) {
            const uint32_t t = a;
            a = b;
            b = t;
        }
        if (atomicCAS(&parent[a], a, b) == a) return;  // hook the larger root under the smaller one
    }
}

// One tree level for one unplaced point i (register state): read the contested child slot, note
// the ancestor just passed, then either settle or descend and bid for the child's free slot.
// Slots of depth L+1 are only touched by points that reached depth L+1, so bidding right away
// cannot disturb a point that is still reading its depth-L slot.  Returns true while unplaced.
template <typename G>
__device__ __forceinline__ bool kd_level_settle(uint32_t *slot, uint32_t i, typename G::Store self, uint32_t child,
                                                typename G::Store pc, uint32_t &st, uint32_t &lox, uint32_t &loy,
                                                uint32_t &hix, uint32_t &hiy);

template <typename G>
__device__ __forceinline__ bool kd_level_step(const typename G::Store *P, uint32_t *slot, uint32_t i,
                                              typename G::Store self, uint32_t &st, uint32_t &lox, uint32_t &loy,
                                              uint32_t &hix, uint32_t &hiy) {
    using R = IdxBits<uint32_t>;
    const uint32_t child = slot[2 * (st & R::MASK) + ((st & R::SIDE) ? 1u : 0u)];
    return kd_level_settle<G>(slot, i, self, child, P[child], st, lox, loy, hix, hiy);
}

template <typename G>
__device__ __forceinline__ bool kd_level_settle(uint32_t *slot, uint32_t i, typename G::Store self, uint32_t child,
                                                typename G::Store pc, uint32_t &st, uint32_t &lox, uint32_t &loy,
                                                uint32_t &hix, uint32_t &hiy) {
    using R = IdxBits<uint32_t>;
    const uint32_t a = st & R::MASK;
    const uint32_t d = (st & R::DIR) ? 1u : 0u;
    // deepest ancestor wins (design/03_dbscan.md); value selects keep the four ids in registers
    const bool right = (st & R::SIDE) != 0;
    lox = (right && !d) ? a : lox;
    loy = (right && d) ? a : loy;
    hix = (!right && !d) ? a : hix;
    hiy = (!right && d) ? a : hiy;
    if (child == i) {
        st = R::PLACED;
        return false;
    }
    const uint32_t nd = d ^ 1u;
    const uint32_t ns = ((nd ? G::sy(self) : G::sx(self)) < (nd ? G::sy(pc) : G::sx(pc))) ? 0u : 1u;
    atomicMin(&slot[2 * child + ns], i);
    st = child | (nd ? R::DIR : 0u) | (ns ? R::SIDE : 0u);
    return true;
}

// The whole pipeline for one segment.  Preconditions: wk.p holds the n points (pid order, already
// in G's storage format), src is the same data in global memory, all threads of the block call
// this with identical arguments, n >= 1, minpts >= 1, eps > 0 finite.  PPT = points per thread
// held in registers during B (LDS tiers, PPT*T >= n) or 0 for the in-memory variant.
// out_labels: global, [n].  Returns the cluster count (valid on every thread).
template <int T, bool GLOBAL, typename Idx, int PPT, typename G>
__device__ __forceinline__ uint32_t dbscan_segment(const DbWork<Idx, G> wk, const double2 *__restrict__ src,
                                                   uint32_t n, double eps, uint32_t minpts, uint32_t nb_log,
                                                   int32_t *out_labels) {
    using B = IdxBits<Idx>;
    using Store = typename G::Store;
    using S = typename G::S;
    const uint32_t tid = threadIdx.x;
    const Store *const P = wk.p;
    Store *const CS = wk.cs;
    uint32_t *const slot = wk.slot;
    Idx *const anc = wk.anc;
    Idx *const pid_s = wk.pid_s;
    Idx *const inv = wk.inv;
    uint8_t *const sflags = wk.sflags;
    uint32_t *const red = wk.red;
    uint32_t *const edges = wk.edges;
    uint32_t *const n_edges = wk.red + 36;
    uint32_t *const anyf = wk.red + 40;  // block_any flags
    uint32_t any_round = 0;
    G geo;
    geo.init(eps);

#ifdef ECAL_PHASE_PROF
    unsigned long long phase_t__ = __builtin_readcyclecounter();
    uint32_t levels__ = 0, sweeps__ = 0;
#endif
    // ---------------- B: kd-cell bounds by level-synchronous insertion replay ----------------
    // Every unplaced point hangs under a node a (split direction d) on one side; the free child
    // slot (a, side) goes to the smallest pid bidding for it (= the next one kd_insert would put
    // there); the others descend to that child.  One barrier per tree level.
    const double tol = eps * 9.313225746154785e-10;  // eps * 2^-30, see point_flags
    uint32_t myflags[PPT > 0 ? PPT : 1];
    for (uint32_t i = tid; i < n; i += T) {
        slot[2 * i] = NONE32;
        slot[2 * i + 1] = NONE32;
    }
    if (tid == 0) {
        *n_edges = 0;
        anyf[0] = anyf[1] = anyf[2] = 0;
    }
    if constexpr (PPT > 0) {
        // LDS tiers: the per-point state lives in registers (points tid, tid+T, ...)
        using R = IdxBits<uint32_t>;
        uint32_t st[PPT], lox[PPT], loy[PPT], hix[PPT], hiy[PPT];
        Store pp[PPT];
#pragma unroll
        for (int u = 0; u < PPT; u++) {
            lox[u] = loy[u] = hix[u] = hiy[u] = B::NONE;
            st[u] = R::PLACED;
        }
        __syncthreads();
        const S x0 = G::sx(P[0]);
#pragma unroll
        for (int u = 0; u < PPT; u++) pp[u] = P[(tid + u * T < n) ? tid + u * T : 0];
        // B.0 wave 0 alone replays the first KTOP insertions (no block barrier: one wave's LDS
        // operations execute in order).  All bids of the top levels would otherwise hit the same
        // two or four LDS words.
        if (tid < KTOP) {
            const uint32_t i = tid;
            if (i > 0 && i < n) {
                const uint32_t side = (G::sx(pp[0]) < x0) ? 0u : 1u;  // kdtree.cpp:128 — ties go right
                atomicMin(&slot[side], i);
                st[0] = side ? R::SIDE : 0u;  // under node 0, dir 0
            }
            for (;;) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                bool act = false;
                if (!(st[0] & R::PLACED)) act = kd_level_step<G>(P, slot, i, pp[0], st[0], lox[0], loy[0], hix[0], hiy[0]);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (!__any(act)) break;
            }
        }
        __syncthreads();
        ECAL_PHASE_MARK(12);  // init + B.0
        // B.1 every later point walks down the finished top tree (reads only) to its first free slot
#pragma unroll
        for (int u = 0; u < PPT; u++) {
            const uint32_t i = tid + u * T;
            if (i < n && i >= KTOP) {
                uint32_t a = 0, d = 0;
                uint32_t side = (G::sx(pp[u]) < x0) ? 0u : 1u;
                for (;;) {
                    const uint32_t child = slot[2 * a + side];
                    if (child == NONE32) break;
                    lox[u] = (side && !d) ? a : lox[u];
                    loy[u] = (side && d) ? a : loy[u];
                    hix[u] = (!side && !d) ? a : hix[u];
                    hiy[u] = (!side && d) ? a : hiy[u];
                    const Store pc = P[child];
                    a = child;
                    d ^= 1u;
                    side = ((d ? G::sy(pp[u]) : G::sx(pp[u])) < (d ? G::sy(pc) : G::sx(pc))) ? 0u : 1u;
                }
                st[u] = a | (d ? R::DIR : 0u) | (side ? R::SIDE : 0u);
            }
        }
        __syncthreads();  // every walk is done before the first bid changes a slot
#pragma unroll
        for (int u = 0; u < PPT; u++) {
            const uint32_t i = tid + u * T;
            if (i < n && i >= KTOP) atomicMin(&slot[2 * (st[u] & R::MASK) + ((st[u] & R::SIDE) ? 1u : 0u)], i);
        }
        __syncthreads();
        ECAL_PHASE_MARK(13);  // B.1 walk + first bids
        // B.2 level-synchronous bidding below the top tree
        for (;;) {
            bool active = false;
#pragma unroll
            for (int u = 0; u < PPT; u++)
                if (!(st[u] & R::PLACED))
                    active |= kd_level_step<G>(P, slot, tid + u * T, pp[u], st[u], lox[u], loy[u], hix[u], hiy[u]);
#ifdef ECAL_PHASE_PROF
            levels__++;
#endif
            if (!block_any(active, anyf, any_round)) break;
        }
        ECAL_PHASE_MARK(14);  // B.2
#pragma unroll
        for (int u = 0; u < PPT; u++) {
            const uint32_t i = tid + u * T;
            myflags[u] = 0;
            if (i < n) {
                anc[4 * i + 0] = (Idx) lox[u];
                anc[4 * i + 1] = (Idx) loy[u];
                anc[4 * i + 2] = (Idx) hix[u];
                anc[4 * i + 3] = (Idx) hiy[u];
                myflags[u] = point_flags<G, Idx>(P, pp[u], lox[u], loy[u], hix[u], hiy[u], tol);
            }
        }
    } else {
        // global tier: same replay with the per-point state in memory.
        // cursor word of an unplaced point: node it hangs under | split dir of that node | side taken
        Idx *const cur = pid_s;
        for (uint32_t i = tid; i < n; i += T) {
            anc[4 * i + 0] = (Idx) B::NONE;
            anc[4 * i + 1] = (Idx) B::NONE;
            anc[4 * i + 2] = (Idx) B::NONE;
            anc[4 * i + 3] = (Idx) B::NONE;
        }
        __syncthreads();
        {
            const S x0 = G::sx(P[0]);
            for (uint32_t i = tid; i < n; i += T) {
                if (i == 0) {
                    cur[0] = (Idx) B::PLACED;
                } else {
                    const uint32_t side = (G::sx(P[i]) < x0) ? 0u : 1u;
                    atomicMin(&slot[side], i);
                    cur[i] = (Idx) (side ? B::SIDE : 0u);
                }
            }
        }
        __syncthreads();
        for (;;) {
            bool active = false;
            for (uint32_t i = tid; i < n; i += T) {
                const uint32_t c = cur[i];
                if (c & B::PLACED) continue;
                const uint32_t a = c & B::MASK;
                const uint32_t d = (c & B::DIR) ? 1u : 0u;
                const uint32_t side = (c & B::SIDE) ? 1u : 0u;
                const uint32_t child = ld_shared_word<GLOBAL>(&slot[2 * a + side]);
                anc[4 * i + (side ? 0 : 2) + d] = (Idx) a;
                if (child == i) {
                    cur[i] = (Idx) B::PLACED;
                } else {
                    const Store pi = P[i], pc = P[child];
                    const uint32_t nd = d ^ 1u;
                    const uint32_t ns = ((nd ? G::sy(pi) : G::sx(pi)) < (nd ? G::sy(pc) : G::sx(pc))) ? 0u : 1u;
                    atomicMin(&slot[2 * child + ns], i);
                    cur[i] = (Idx) (child | (nd ? B::DIR : 0u) | (ns ? B::SIDE : 0u));
                    active = true;
                }
            }
#ifdef ECAL_PHASE_PROF
            levels__++;
#endif
            if (!block_any(active, anyf, any_round)) break;
        }
        for (uint32_t i = tid; i < n; i += T)
            wk.pflags[i] = (uint8_t) point_flags<G, Idx>(P, P[i], anc[4 * i], anc[4 * i + 1], anc[4 * i + 2],
                                                         anc[4 * i + 3], tol);
    }
    ECAL_PHASE_MARK(0);
    ECAL_PHASE_COUNT(8, levels__);

    uint32_t *const parent = slot;       // [n]  D/E: union-find parent by pid (NONE32 = not core)
    uint32_t *const arr = slot + n + 1;  // [nb + 2]
    uint32_t *const bstart = arr + 1;    // bstart[b] = first bucket position of bucket b; bstart[nb] = n
    const uint32_t nb = 1u << nb_log;
    Grid g;
    g.bx_log = (nb_log + 1) >> 1;
    g.bx_mask = (1u << g.bx_log) - 1u;
    g.by_mask = (1u << (nb_log - g.bx_log)) - 1u;
    bool use_grid = false;
    {
    // ---------------- C: torus cell hash + counting sort ----------------
    bool bad = !geo.usable();
    for (uint32_t b = tid; b < nb + 2; b += T) arr[b] = 0;
    for (uint32_t i = tid; i < n; i += T) {
        const Store p = P[i];
        if (!geo.cell_ok(G::sx(p)) || !geo.cell_ok(G::sy(p))) bad = true;  // also catches NaN/inf
    }
    use_grid = !block_any(bad, anyf, any_round);
    if (use_grid) {
        for (uint32_t i = tid; i < n; i += T) {
            const Store p = P[i];
            atomicAdd(&arr[1 + g.bucket(geo.cell(G::sx(p)), geo.cell(G::sy(p)))], 1u);
        }
        __syncthreads();
        // inclusive scan of the counts: arr[1+b] = end of bucket b (blocked: nb/T buckets per thread)
        {
            const uint32_t per = (nb + T - 1) / T;
            const uint32_t b0 = tid * per;
            uint32_t sum = 0;
            for (uint32_t b = b0; b < b0 + per && b < nb; b++) sum += ld_shared_word<GLOBAL>(&arr[1 + b]);
            uint32_t total;
            uint32_t run = block_exscan<T>(sum, red, &total);
            for (uint32_t b = b0; b < b0 + per && b < nb; b++) {
                run += ld_shared_word<GLOBAL>(&arr[1 + b]);
                arr[1 + b] = run;
            }
            if (tid == 0) arr[1 + nb] = n;
        }
        __syncthreads();
        // scatter from the end of each bucket; afterwards arr[1+b] = bstart[b] = start of bucket b.
        // LDS tiers: CS aliases P, so the coordinates are re-read from global memory (L2-hot).
        auto place = [&](uint32_t i, uint32_t fl) {
            const Store p = G::pack(src[i]);
            const uint32_t pos = atomicSub(&arr[1 + g.bucket(geo.cell(G::sx(p)), geo.cell(G::sy(p)))], 1u) - 1u;
            CS[pos] = p;
            pid_s[pos] = (Idx) i;
            inv[i] = (Idx) pos;
            sflags[pos] = (uint8_t) fl;
        };
        if constexpr (PPT > 0) {
#pragma unroll
            for (int u = 0; u < PPT; u++)
                if (tid + u * T < n) place(tid + u * T, myflags[u]);
        } else {
            for (uint32_t i = tid; i < n; i += T) place(i, wk.pflags[i]);
        }
    } else {
        auto place = [&](uint32_t i, uint32_t fl) {
            CS[i] = G::pack(src[i]);
            pid_s[i] = (Idx) i;
            inv[i] = (Idx) i;
            sflags[i] = (uint8_t) fl;
        };
        if constexpr (PPT > 0) {
#pragma unroll
            for (int u = 0; u < PPT; u++)
                if (tid + u * T < n) place(tid + u * T, myflags[u]);
        } else {
            for (uint32_t i = tid; i < n; i += T) place(i, wk.pflags[i]);
        }
    }
    __syncthreads();
    ECAL_PHASE_MARK(1);

    // ---------------- D: neighbour counts (stopped at minpts) -> core flags ----------------
    // (The kernel is instruction-issue bound, not latency bound — profiles/r01_notes.md — so the
    // scans run one query at a time with early exit rather than several in lockstep.)
    for (uint32_t kq = tid; kq < n; kq += T) {
        const Store qs = CS[kq];
        const S qx = G::sx(qs), qy = G::sy(qs);
        const uint32_t pi = pid_s[kq];
        uint32_t cnt = 0;
        for_candidates<GLOBAL>(geo, CS, sflags, n, use_grid, g, bstart, qx, qy,
                               [&](uint32_t k, Store c, uint32_t fk) -> bool {
                                   const S dx = G::sx(c) - qx, dy = G::sy(c) - qy;  // node.pos - query.pos
                                   if (geo.in_ball(dx, dy) && k != kq) {
                                       bool pr = false;
                                       if (maybe_pruned(geo, dx, dy, fk)) {  // rare: exact replay of the pruning
                                           const uint32_t pj = pid_s[k];
                                           const double2 q = G::unpack(qs), cd = G::unpack(c);
                                           const double aeps = fabs(eps);
                                           if (fabs(cd.x - q.x) == aeps) pr = pruned_dim<G>(CS, inv, anc, pj, 0u, q.x, eps);
                                           if (!pr && fabs(cd.y - q.y) == aeps)
                                               pr = pruned_dim<G>(CS, inv, anc, pj, 1u, q.y, eps);
                                       }
                                       if (!pr) cnt++;
                                   }
                                   return cnt < minpts;
                               });
        parent[pi] = (cnt >= minpts) ? pi : NONE32;
        if (cnt >= minpts) sflags[kq] |= 16u;  // own byte: core marker for the E sweep
    }
    __syncthreads();
    ECAL_PHASE_MARK(2);

    // ---------------- E: min-seed directed reachability over core points ----------------
    // E.1 one sweep: union the two-way edges, remember the one-way ones
    for (uint32_t kq = tid; kq < n; kq += T) {
        const uint32_t fi = sflags[kq];
        if (!(fi & 16u)) continue;
        const uint32_t pi = pid_s[kq];
        const Store qs = CS[kq];
        const S qx = G::sx(qs), qy = G::sy(qs);
        for_candidates<GLOBAL>(
            geo, CS, sflags, n, use_grid, g, bstart, qx, qy, [&](uint32_t k, Store c, uint32_t fk) -> bool {
                const S dx = G::sx(c) - qx, dy = G::sy(c) - qy;
                if (geo.in_ball(dx, dy) && k != kq && (fk & 16u)) {
                    if (maybe_pruned(geo, dx, dy, fk) || maybe_pruned(geo, -dx, -dy, fi)) {
                        // rare: decide both directions exactly
                        const uint32_t pj = pid_s[k];
                        bool fwd, bwd;  // edge i->j, edge j->i
                        edge_dirs<G>(CS, inv, anc, pi, pj, G::unpack(qs), G::unpack(c), eps, fwd, bwd);
                        if (fwd && bwd) {
                            if (k < kq) uf_union<GLOBAL>(parent, pi, pj);
                        } else if (fwd) {  // the j side sees (!fwd, bwd) and skips: each one-way edge once
                            const uint32_t at = atomicAdd(n_edges, 1u);
                            if (at < EDGE_CAP) {
                                edges[2 * at] = pi;
                                edges[2 * at + 1] = pj;
                            }
                        }
                    } else if (k < kq) {  // two-way edge; each pair once
                        uf_union<GLOBAL>(parent, pi, pid_s[k]);
                    }
                }
                return true;
            });
    }
    }
    __syncthreads();
    const uint32_t m_edges = *n_edges;
    if (m_edges <= EDGE_CAP) {
        // E.2 flatten: parent[p] = root of p's two-way component (= its smallest pid)
        for (uint32_t i = tid; i < n; i += T) {
            if (ld_shared_word<GLOBAL>(&parent[i]) != NONE32) parent[i] = uf_root<GLOBAL>(parent, i);
        }
        __syncthreads();
        if (m_edges > 0) {
            // E.3 a one-way edge u->v lowers v's component to u's label; iterate to the fix-point.
            // anc is dead from here on: it holds the component labels (u32 per pid).
            uint32_t *const comp = reinterpret_cast<uint32_t *>(anc);
            for (uint32_t i = tid; i < n; i += T) comp[i] = i;
            __syncthreads();
            for (;;) {
                bool changed = false;
                for (uint32_t e = tid; e < m_edges; e += T) {
                    const uint32_t ru = ld_shared_word<GLOBAL>(&parent[edges[2 * e]]);
                    const uint32_t rv = ld_shared_word<GLOBAL>(&parent[edges[2 * e + 1]]);
                    const uint32_t lu = ld_shared_word<GLOBAL>(&comp[ru]);
                    if (lu < ld_shared_word<GLOBAL>(&comp[rv])) {
                        atomicMin(&comp[rv], lu);
                        changed = true;
                    }
                }
                if (!block_any(changed, anyf, any_round)) break;
            }
            // final label of p = comp[root(p)] (every thread touches parent[] only at its own pids)
            for (uint32_t i = tid; i < n; i += T) {
                const uint32_t r = ld_shared_word<GLOBAL>(&parent[i]);
                if (r != NONE32) parent[i] = ld_shared_word<GLOBAL>(&comp[r]);
            }
            __syncthreads();
        }
    } else {
        // Fallback (more one-way edges than the list holds): push/pull sweeps + pointer jumping on
        // label[] until a fix-point — the general form of E, independent of the union-find state.
        uint32_t *const label = parent;
        for (uint32_t i = tid; i < n; i += T) {
            if (ld_shared_word<GLOBAL>(&label[i]) != NONE32) label[i] = i;
        }
        __syncthreads();
        for (;;) {
            bool changed = false;
#ifdef ECAL_PHASE_PROF
            sweeps__++;
#endif
            for (uint32_t kq = tid; kq < n; kq += T) {
                const uint32_t pi = pid_s[kq];
                const uint32_t L0 = ld_shared_word<GLOBAL>(&label[pi]);
                if (L0 == NONE32) continue;
                uint32_t L = L0;
                const Store qs = CS[kq];
                const S qx = G::sx(qs), qy = G::sy(qs);
                for_candidates<GLOBAL>(geo, CS, sflags, n, use_grid, g, bstart, qx, qy,
                                       [&](uint32_t k, Store c, uint32_t) -> bool {
                                           const S dx = G::sx(c) - qx, dy = G::sy(c) - qy;
                                           if (geo.in_ball(dx, dy) && k != kq) {
                                               const uint32_t pj = pid_s[k];
                                               const uint32_t Lj = ld_shared_word<GLOBAL>(&label[pj]);
                                               if (Lj == NONE32) return true;
                                               bool fwd, bwd;
                                               edge_dirs<G>(CS, inv, anc, pi, pj, G::unpack(qs), G::unpack(c), eps,
                                                            fwd, bwd);
                                               if (bwd && Lj < L) L = Lj;
                                               if (fwd && L < Lj) {
                                                   atomicMin(&label[pj], L);
                                                   changed = true;
                                               }
                                           }
                                           return true;
                                       });
                if (L < L0) {
                    atomicMin(&label[pi], L);
                    changed = true;
                }
            }
            __syncthreads();
            // pointer jumping: label[v] always names a core point that reaches v, so does label[label[v]]
            for (uint32_t i = tid; i < n; i += T) {
                uint32_t L = ld_shared_word<GLOBAL>(&label[i]);
                if (L == NONE32) continue;
                const uint32_t L0 = L;
                for (;;) {
                    const uint32_t up = ld_shared_word<GLOBAL>(&label[L]);
                    if (up >= L) break;
                    L = up;
                }
                if (L < L0) atomicMin(&label[i], L);
            }
            if (!block_any(changed, anyf, any_round)) break;
        }
    }
    ECAL_PHASE_MARK(3);
    ECAL_PHASE_COUNT(9, sweeps__);
    ECAL_PHASE_COUNT(10, 1);
    ECAL_PHASE_COUNT(11, m_edges);

    // ---------------- F: seeds ranked in pid order = reference cluster ids ----------------
    uint32_t *const label = parent;
    Idx *const rank = pid_s;
    uint32_t total;
    {
        const uint32_t per = (n + T - 1) / T;
        const uint32_t i0 = tid * per;
        uint32_t mine = 0;
        for (uint32_t i = i0; i < i0 + per && i < n; i++) mine += (ld_shared_word<GLOBAL>(&label[i]) == i) ? 1u : 0u;
        uint32_t run = block_exscan<T>(mine, red, &total);
        for (uint32_t i = i0; i < i0 + per && i < n; i++) {
            if (ld_shared_word<GLOBAL>(&label[i]) == i) rank[i] = (Idx) (run++);
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < n; i += T) {
        const uint32_t L = ld_shared_word<GLOBAL>(&label[i]);
        out_labels[i] = (L == NONE32) ? -1 : (int32_t) rank[L];
    }
    __syncthreads();
    ECAL_PHASE_MARK(4);
    return total;
}

}  // namespace ecal
