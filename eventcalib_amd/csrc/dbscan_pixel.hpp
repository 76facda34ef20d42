// DBSCAN of one segment of event PIXELS (integer coordinates) by one workgroup, in 22 KB of LDS (seven per CU).
//
// Same contract as dbscan_device.hpp (labels identical to dbscan/include/dbscan.h:115-265 on top of
// dbscan/src/kdtree.cpp:106-179, including the range query's pruning quirk); this is the form the hot path
// takes: segments of <= 768 points whose coordinates are integers (|v| <= 16383), whose bounding box padded
// by the disc radius fits a 3232-word bitmap (a 346x260 sensor does) and eps < 16.  Anything else is appended
// to a to-do list that the general kernels work off.
//
// What makes it small and why that matters: the kernel is latency bound — a workgroup is a ~45 us chain of
// dependent LDS operations and barriers, and kernel time is ~ 1 / (workgroups per CU) from 1 to 6
// (tools/occupancy_probe.sh, profiles/r01_notes.md) — so throughput is proportional to the workgroups a CU can
// hold, i.e. to 160 KB / LDS per workgroup (capacity 768 points instead of 1024: 22.1 KB, 7 instead of 6 workgroups,
// 1.157 -> 1.033 ms on the benchmark stream whose segments hold <= 666 points).  For integer pixels the
// closed form of the pruning quirk (design/03_dbscan.md) collapses to two bits per point,
//     f_d(j) = "an ancestor of j in the insertion-order kd-tree splits on d at j's own d-coordinate",
// and  i -> j is pruned  <=>  j = i + eps * e_d exactly  and  f_d(j)   (integral eps only),
// so the replay of kd_insert (phase B) only has to produce those bits: no ancestor tables, no f64 replay.
// LDS regions are reused across phases (points -> rank tables, child slots -> bitmap).
//
// Phases: A load/pack + bounding box; B kd replay -> f bits (wave 0 builds the first 64 nodes, the rest walk
// that top tree read-only, then level-synchronous bidding with one barrier per level); bitmap of the points;
// raster ranks (row starts + per-word prefix counts); D core test = popcount of the disc's row windows;
// E union-find over the set bits of the half disc (each pair once), one-way edges to a list + fix-point;
// F seeds ranked in pid order = the reference's cluster ids.
#pragma once
#include <type_traits>
#include "dbscan_device.hpp"

namespace ecal {

#ifndef ECAL_PX_T
#define ECAL_PX_T 256
#endif
constexpr int PX_T = ECAL_PX_T;
constexpr int PX_CAP = 768;    // points per segment of the first pass (three per thread)
constexpr int PX_CAP2 = 2048;  // second pass over the segments the first one left: eight per thread, 10-bit bitmap coordinates
constexpr uint32_t PX_WORDS = 3232;    // bitmap words: 346x260 padded by 2*4 = 354x268 bits = 268 rows x 12 words = 3216
constexpr uint32_t PX_ROWS = 472;      // rowstart[PX_ROWS + 1]: 512 u16
constexpr uint32_t PX_EDGE_CAP = 64;   // one-way edges kept (more: left to the general kernel)
constexpr int PX_RMAX = 15;

// Everything the kernel needs to know about eps, worked out once on the host (the f64 square roots used to run in
// every thread).  dmask[k] = row dy = k - Rd of the closed eps-disc as bits of the (2 Rd + 1)-wide window that starts
// at column cx - Rd: one 64-bit fetch per row serves every row of the disc, the row's shape is an AND.
struct PxGeom {
    int e2i;           // floor(eps^2): d2 <= eps^2  <=>  d2 <= e2i for integer deltas
    int Rd;            // floor(eps) = the disc's radius in pixels, <= PX_RMAX
    uint32_t eps_int;  // eps is integral (== Rd): only then |delta| == eps exists and the pruning quirk can bite
    uint32_t dmask[2 * PX_RMAX + 1];
    // the half disc (rows above + own row to the left) packed row after row into one word (radius <= 4: 24 bits):
    // hd_code[p] = k << 5 | b of packed position p (k = dy + Rd, b = column - (cx - Rd))
    uint8_t hd_code[32];
    // hd_cover[p] = the packed positions within a two-way edge of position p (d2 <= floor(eps^2), minus the exactly-eps
    // pairs when eps is integral): once p is joined, these need no union of their own
    uint32_t hd_cover[32];
};

__host__ __device__ constexpr int px_isqrt(int v) {
    int r = 0;
    while ((long long) (r + 1) * (r + 1) <= (long long) v) r++;
    return r;
}
// half-disc row k as a field of its window: lowest column, width, position in the packed word
__host__ __device__ constexpr int px_hd_low(int e2i, int Rd, int k) { return k == Rd ? 0 : Rd - px_isqrt(e2i - (k - Rd) * (k - Rd)); }
__host__ __device__ constexpr int px_hd_width(int e2i, int Rd, int k) {
    return k == Rd ? Rd : 2 * px_isqrt(e2i - (k - Rd) * (k - Rd)) + 1;
}
__host__ __device__ constexpr int px_hd_pos(int e2i, int Rd, int k) {
    int p = 0;
    for (int j = 0; j < k; j++) p += px_hd_width(e2i, Rd, j);
    return p;
}
// nibble |dy| of the result = 1 + the largest |dx| with dx^2 + dy^2 <= lim (0: none), for |dy| <= 7
__host__ __device__ constexpr uint32_t px_reach_table(int lim) {
    uint32_t c = 0;
    for (int dy = 0; dy < 8; dy++) {
        const int r = lim - dy * dy;
        const uint32_t v = r < 0 ? 0u : (uint32_t) (px_isqrt(r) + 1);
        c |= (v > 15u ? 15u : v) << (4 * dy);
    }
    return c;
}
__host__ __device__ constexpr uint32_t px_disc_mask(int e2i, int Rd, int k) {
    const int dy = k - Rd;
    const int w = px_isqrt(e2i - dy * dy);
    return ((w >= 15 ? 0x7FFFFFFFu : ((1u << (2 * w + 1)) - 1u)) << (Rd - w));
}
// false: eps is outside what the pixel kernel represents (the general tiers take the launch)
inline bool px_geometry(double eps, PxGeom *g) {
    if (!(eps > 0.0) || !(eps < (double) (PX_RMAX + 1))) return false;
    const double e2 = eps * eps;  // GeoI16::init
    g->e2i = (int) floor(e2);
    g->Rd = px_isqrt(g->e2i);
    g->eps_int = (eps == floor(eps)) ? 1u : 0u;
    for (int k = 0; k < 2 * PX_RMAX + 1; k++) g->dmask[k] = (k <= 2 * g->Rd) ? px_disc_mask(g->e2i, g->Rd, k) : 0u;
    for (int p = 0; p < 32; p++) {
        g->hd_code[p] = 0;
        g->hd_cover[p] = 0;
    }
    int np = 0;  // packed positions
    if (g->Rd <= 4 && px_hd_pos(g->e2i, g->Rd, g->Rd + 1) <= 32)  // only a half disc that fits one word is ever packed
        for (int k = 0; k <= g->Rd; k++)
            for (int b = px_hd_low(g->e2i, g->Rd, k); b < px_hd_low(g->e2i, g->Rd, k) + px_hd_width(g->e2i, g->Rd, k); b++)
                g->hd_code[np++] = (uint8_t) (k << 5 | b);
    const int tlim = g->eps_int ? g->e2i - 1 : g->e2i;
    for (int p = 0; p < np; p++)
        for (int q = 0; q < np; q++) {
            const int dx = (g->hd_code[p] & 31) - (g->hd_code[q] & 31), dy = (g->hd_code[p] >> 5) - (g->hd_code[q] >> 5);
            if (dx * dx + dy * dy <= tlim) g->hd_cover[p] |= 1u << q;
        }
    return g->Rd >= 0 && g->Rd <= PX_RMAX;
}

template <int CAP>
struct PixelLayout {
    // region A (from the rank phase on): rank -> pid | f << PB | core << (PB + 2), the one-way edge list, the disc masks
    static constexpr size_t pf_off = 0;                                     // u16[CAP]
    static constexpr size_t edges_off = pf_off + 2 * CAP;                   // u32[2 * PX_EDGE_CAP]
    static constexpr size_t dm_off = edges_off + 8 * PX_EDGE_CAP;           // u32[32]
    static constexpr size_t hd_off = dm_off + 128;                          // u8[32]
    // region B: kd child slots during B (+ one dummy word that stays NONE) — they may run on into parent[], which
    // phase D is the first to use; then the bitmap; after E.1: component labels
    static constexpr size_t slot_off = hd_off + 32;                         // u32[2 * CAP + 1]
    static constexpr size_t bm_off = slot_off;                              // u32[PX_WORDS]
    static constexpr size_t parent_off = bm_off + 4 * PX_WORDS + 16;        // u32[CAP] (after one spare bitmap word)
    static_assert(4 * PX_WORDS + 16 + 4 * CAP >= 8 * CAP + 4, "child slots must fit bitmap + parent");
    static_assert(slot_off % 16 == 0 && PX_WORDS % 4 == 0, "the bitmap is cleared with 16-byte stores");
    static_assert(PX_WORDS >= (uint32_t) CAP, "component labels (E.3) must fit the bitmap region");
    static constexpr size_t rowstart_off = parent_off + 4 * CAP;            // u16[512]
    static constexpr size_t wpre_off = rowstart_off + 1024;                 // u8[PX_WORDS]
    static constexpr size_t red_off = wpre_off + PX_WORDS;                  // u32[48]
    static constexpr size_t bytes = red_off + 4 * 48;
};

// Word formats by capacity.  Child-slot word: pid << 2 CB | x' << CB | y' (x', y' = bitmap coordinates < 2^CB - 1):
// ds_min_u32 orders the bids by pid, and the winner's coordinates come back with its id — one dependent LDS read per
// tree level instead of two.  Rank table entry (u16): pid | f << PB | core << (PB + 2).
template <int CAP>
struct PxFmt {
    static constexpr uint32_t PB = CAP > 1024 ? 11u : 10u;   // pid bits
    static constexpr uint32_t CB = (32u - PB) / 2u;          // coordinate bits: 11 (CAP 1024) or 10 (CAP 2048)
    static constexpr uint32_t CMASK = (1u << CB) - 1u;
    static constexpr uint32_t PF_PID = (1u << PB) - 1u, PF_CORE = 1u << (PB + 2u);
    static constexpr uint32_t DUMMY_SLOT = 8u * CAP;         // byte offset of the child-slot word nobody bids for
    static __device__ __forceinline__ uint32_t word(uint32_t pid, uint32_t cx, uint32_t cy) { return (pid << (2u * CB)) | (cx << CB) | cy; }
    static __device__ __forceinline__ uint32_t wx(uint32_t w) { return (w >> CB) & CMASK; }
    static __device__ __forceinline__ uint32_t wy(uint32_t w) { return w & CMASK; }
};

// One tree level for one unplaced point (register state).  sl = byte offset of the child slot the point bid for
// (~0 = placed), sh = bit offset of the coordinate the winner of that slot splits on (CB = x, 0 = y), fb = the prune
// bit of that dimension.  Returns true while unplaced.
template <int CAP>
__device__ __forceinline__ bool px_level_step(unsigned char *slotb, uint32_t i, uint32_t self, uint32_t &sl, uint32_t &sh,
                                              uint32_t &fb, uint32_t &f) {
    using F = PxFmt<CAP>;
    const uint32_t cw = *reinterpret_cast<const uint32_t *>(slotb + sl);
    const uint32_t child = cw >> (2u * F::CB);
    if (child == i) {
        sl = ~0u;
        return false;
    }
    const uint32_t sv = __builtin_amdgcn_ubfe(self, sh, F::CB), cv = __builtin_amdgcn_ubfe(cw, sh, F::CB);
    f |= (sv == cv) ? fb : 0u;                    // `child` becomes an ancestor splitting at my coordinate
    sl = (child << 3) | (sv < cv ? 0u : 4u);      // kdtree.cpp:128-131: left iff strictly smaller
    atomicMin(reinterpret_cast<uint32_t *>(slotb + sl), self);
    sh ^= F::CB;
    fb ^= 3u;
    return true;
}

// returns the number of clusters, or -1 when the segment was left to the to-do list.  KNOWN: the segment's count and offset are handed in (the fused
// pass, ecal_fused.hip: they were written by this very workgroup a moment ago, and a scalar load could find a stale line of
// the constant cache, which a neighbouring workgroup may have filled with the array's previous contents)
template <int E2I, int CAP, bool KNOWN = false>
__device__ __forceinline__ int px_segment(unsigned char *px_smem, const uint32_t s, const double *__restrict__ xy,
                                           const uint32_t *__restrict__ seg_off, const uint32_t *__restrict__ seg_cnt,
                                           const PxGeom &geom, uint32_t minpts, int32_t *__restrict__ labels,
                                           uint32_t *__restrict__ n_clusters, uint32_t *__restrict__ todo,
                                           uint32_t *__restrict__ todo_count, uint32_t known_cnt = 0, uint32_t known_off = 0,
                                           const uint32_t *__restrict__ xy16 = nullptr, const uint32_t *__restrict__ seg_fmt = nullptr,
                                           uint32_t *__restrict__ tree_out = nullptr, uint32_t *__restrict__ tree_flag = nullptr,
                                           uint32_t tree_epoch = 0) {
    using L = PixelLayout<CAP>;
    using F = PxFmt<CAP>;
    using G = GeoI16;
    constexpr int T = PX_T, PPT = CAP / PX_T;
    const uint32_t tid = threadIdx.x;
#ifdef ECAL_PHASE_PROF
    unsigned long long phase_t__ = __builtin_readcyclecounter(), d7__ = 0;
    uint32_t levels__ = 0;
#endif
    const uint32_t n = KNOWN ? known_cnt : seg_cnt[s];
    const uint32_t base32 = KNOWN ? known_off : seg_off[s];  // asked for together with the count: one trip to memory instead of two in a row
    asm volatile("" ::"s"(base32));     // (a use right here, or the compiler sinks the load below the branches on n)
    if (n == 0) {
        if (tid == 0) n_clusters[s] = 0;
        return 0;
    }
    if (n > (uint32_t) CAP) {
        if (tid == 0 && todo) todo[atomicAdd(todo_count, 1u)] = s;   // (todo == nullptr: the caller takes the segment on itself — -1 comes back)
        return -1;
    }
    // E2I > 0: the disc is a compile-time constant (masks become literals, the row loops unroll)
    const int Rd = E2I > 0 ? px_isqrt(E2I) : geom.Rd;
    const int e2i = E2I > 0 ? E2I : geom.e2i;
    const bool eps_int = geom.eps_int != 0;  // then eps == Rd
    uint16_t *const pf = reinterpret_cast<uint16_t *>(px_smem + L::pf_off);
    uint32_t *const edges = reinterpret_cast<uint32_t *>(px_smem + L::edges_off);
    uint32_t *const dm = reinterpret_cast<uint32_t *>(px_smem + L::dm_off);
    uint8_t *const hd = px_smem + L::hd_off;
    unsigned char *const slotb = px_smem + L::slot_off;
    uint32_t *const slot = reinterpret_cast<uint32_t *>(px_smem + L::slot_off);
    uint32_t *const bm = reinterpret_cast<uint32_t *>(px_smem + L::bm_off);
    uint32_t *const parent = reinterpret_cast<uint32_t *>(px_smem + L::parent_off);
    uint16_t *const rowstart = reinterpret_cast<uint16_t *>(px_smem + L::rowstart_off);
    uint8_t *const wpre = reinterpret_cast<uint8_t *>(px_smem + L::wpre_off);
    uint32_t *const red = reinterpret_cast<uint32_t *>(px_smem + L::red_off);
    uint32_t *const n_edges = red + 36;
    uint32_t *const rootw = red + 37;
    uint32_t *const anyf = red + 40;
    int *const bbox = reinterpret_cast<int *>(red + 44);  // min x, min y, -max x, -max y
    uint32_t any_round = 0;
    const size_t base = base32;
    const double2 *src = reinterpret_cast<const double2 *>(xy) + base;
    // first point of this wave's u-th batch: batches that start at or after n are skipped with a scalar branch
    const uint32_t wbase = __builtin_amdgcn_readfirstlane(tid);
#ifdef ECAL_PHASE_PROF
    if (tid == 0) {  // scalar loads done (count, offset)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long now__ = __builtin_readcyclecounter();
        d7__ = now__ - phase_t__ + (base & 0);  // added to the counters later: an atomic here would sit in vmcnt
        phase_t__ = now__;
    }
#endif
#define PX_BAIL()                                                  \
    do {                                                           \
        if (tid == 0 && todo) todo[atomicAdd(todo_count, 1u)] = s; \
        return -1;                                                \
    } while (0)
    // debug builds (-DECAL_PX_STOP=k, tools/px_stop_probe.sh): leave after phase k with the phase's results written
    // out, so that instruction counters can be attributed to phases
#ifdef ECAL_PX_STOP
#define PX_STOP(k, expr)                                                                  \
    if (ECAL_PX_STOP == (k)) {                                                            \
        _Pragma("unroll") for (int u = 0; u < PPT; u++) {                                 \
            const uint32_t i = tid + u * T;                                               \
            if (i < n) labels[base + i] = (int32_t) (expr);                               \
        }                                                                                 \
        if (tid == 0) n_clusters[s] = 0;                                                  \
        return 0;                                                                         \
    }
#else
#define PX_STOP(k, expr)
#endif
    // ---------------- A: load, pack, bounding box ----------------
    if (tid < 3) anyf[tid] = 0;
    if (tid < 4) bbox[tid] = 0x7FFFFFFF;
    if (tid == 0) {
        *n_edges = 0;
        slot[F::DUMMY_SLOT / 4] = NONE32;
    }
    __syncthreads();
    uint32_t pp[PPT];
    bool fits = true;
    {
        typedef short short2v __attribute__((ext_vector_type(2)));
        short2v mn = {0x7FFF, 0x7FFF}, mx = {-0x7FFF, -0x7FFF};
        // all of the thread's loads are issued before the first is used (index clamped instead of a branch around the
        // load): with the load inside `if (i < n)` the compiler waited for each one in turn — up to four serial HBM
        // round trips at the head of every workgroup
        // (packed points, ecal_packed_points: a segment the slicer wrote as x | y << 16 is read as that — 4 bytes a point
        // instead of 16, nothing to test: it holds sensor pixels)
        const bool packed = xy16 && (seg_fmt[s] & 1u);
        double2 vin[PPT];
        uint32_t win[PPT];
        if (packed) {
#pragma unroll
            for (int u = 0; u < PPT; u++) win[u] = xy16[base + min(tid + u * T, n - 1u)];
        } else {
#pragma unroll
            for (int u = 0; u < PPT; u++) vin[u] = src[min(tid + u * T, n - 1u)];
        }
#pragma unroll
        for (int u = 0; u < PPT; u++) {
            const uint32_t i = tid + u * T;
            pp[u] = 0;
            if (i < n) {
                if (packed) {
                    pp[u] = win[u];
                } else {
                    const double2 v = vin[u];
                    fits = fits && G::fits(v);
                    pp[u] = G::pack(v);
                }
                slot[2 * i] = NONE32;
                slot[2 * i + 1] = NONE32;
                const short2v c = {(short) G::sx(pp[u]), (short) G::sy(pp[u])};
                mn = __builtin_elementwise_min(mn, c);  // v_pk_min_i16: both coordinates at once
                mx = __builtin_elementwise_max(mx, c);
            }
        }
#ifdef ECAL_PHASE_PROF
        if (tid == 0) {  // own points arrived
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long now__ = __builtin_readcyclecounter();
            atomicAdd(&g_phase_cycles[7], d7__);
            atomicAdd(&g_phase_cycles[9], now__ - phase_t__);
            phase_t__ = now__;
        }
#endif
        for (int o = 32; o > 0; o >>= 1) {
            const int a = __shfl_xor(__builtin_bit_cast(int, mn), o, 64), b = __shfl_xor(__builtin_bit_cast(int, mx), o, 64);
            mn = __builtin_elementwise_min(mn, __builtin_bit_cast(short2v, a));
            mx = __builtin_elementwise_max(mx, __builtin_bit_cast(short2v, b));
        }
        if ((tid & 63) == 0 && mn.x != 0x7FFF) {
            atomicMin(&bbox[0], (int) mn.x);
            atomicMin(&bbox[1], (int) mn.y);
            atomicMin(&bbox[2], -(int) mx.x);
            atomicMin(&bbox[3], -(int) mx.y);
        }
    }
    // the disc tables (read from phase D on; here, behind the point loads, their trip to the kernel arguments is free)
    if (E2I == 0 && tid < (uint32_t) (2 * PX_RMAX + 1)) dm[tid] = geom.dmask[tid];
    if (E2I > 0 && tid < 32u) {
        hd[tid] = geom.hd_code[tid];
        dm[tid] = geom.hd_cover[tid];  // (the run-time disc masks are not needed when the disc is compiled in)
    }
    if (block_any(!fits, anyf, any_round)) PX_BAIL();
    if (Rd > PX_RMAX) PX_BAIL();
    const int ox = bbox[0] - Rd, oy = bbox[1] - Rd;
    const uint32_t W = (uint32_t) (-bbox[2] - bbox[0] + 1 + 2 * Rd), H = (uint32_t) (-bbox[3] - bbox[1] + 1 + 2 * Rd);
    // 64-bit window fetches may read the first word of the next row (or the spare word after the last row):
    // those bits are always masked off
    const uint32_t RW = (W + 31u) >> 5;
    if (H > PX_ROWS || (uint64_t) H * RW > PX_WORDS || W > F::CMASK || H > F::CMASK) PX_BAIL();
    uint32_t mcx[PPT], myy[PPT], me[PPT];
#pragma unroll
    for (int u = 0; u < PPT; u++) {
        mcx[u] = (uint32_t) (G::sx(pp[u]) - ox);
        myy[u] = (uint32_t) (G::sy(pp[u]) - oy);
        me[u] = F::word(tid + u * T, mcx[u] & F::CMASK, myy[u] & F::CMASK);
    }
    if (tid == 0) *rootw = me[0];
    PX_STOP(1, me[u] + W + H);

    // ---------------- B: kd_insert replay -> prune bits ----------------
    uint32_t sl[PPT], sh[PPT], fb[PPT], f[PPT];
#pragma unroll
    for (int u = 0; u < PPT; u++) {
        sl[u] = ~0u;
        sh[u] = 0;
        fb[u] = 2;
        f[u] = 0;
    }
    // B.0: wave 0 alone replays the first KTOP insertions (wave-synchronous, no block barrier)
    if (tid < KTOP) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t x0 = F::wx(*rootw);
        const uint32_t i = tid;
        if (i > 0 && i < n) {
            f[0] |= (mcx[0] == x0) ? 1u : 0u;
            sl[0] = (mcx[0] < x0) ? 0u : 4u;  // children of the root split on y: sh = 0, fb = 2
            atomicMin(reinterpret_cast<uint32_t *>(slotb + sl[0]), me[0]);
        }
        for (;;) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            bool act = false;
            if (sl[0] != ~0u) act = px_level_step<CAP>(slotb, i, me[0], sl[0], sh[0], fb[0], f[0]);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (!__any(act)) break;
        }
    }
    __syncthreads();
    ECAL_PHASE_MARK(12);
    // B.1: every later point walks the finished top tree (reads only; nobody writes during B.1).  The 64 walks of a
    // batch (u, wave) start at the root together and go down one level per step, so the splitting dimension is the same
    // for the whole wave (no per-lane selects); a walk that has ended keeps reading its empty slot.  One batch after the
    // other: each loop runs to the depth of ITS deepest walk — interleaving a thread's batches in one loop made every level
    // cost the deepest of all of them, and the kernel is bound by VALU issue, not by these LDS round trips.
    {
        const uint32_t x0 = F::wx(*rootw);
#pragma unroll
        for (int u = 0; u < PPT; u++) {
            if (!(wbase + u * T < n && wbase + u * T + 63u >= KTOP)) continue;   // batch without a point to walk (wave-uniform)
            const uint32_t i = tid + u * T;
            const bool go = i < n && i >= KTOP;
            uint32_t a2 = go ? ((mcx[u] < x0) ? 0u : 4u) : F::DUMMY_SLOT, dl = 0;
            uint32_t fcx = (go && mcx[u] == x0) ? 1u : 0u, fcy = 0;
            for (;;) {  // two levels per turn: y, then x
                uint32_t cw = *reinterpret_cast<const uint32_t *>(slotb + a2);
                bool live = cw != NONE32;
                if (live) {
                    const uint32_t cv = F::wy(cw);
                    fcy += (myy[u] == cv) ? 1u : 0u;
                    a2 = ((cw >> (2u * F::CB)) << 3) | (myy[u] < cv ? 0u : 4u);
                    dl = 1;
                    cw = *reinterpret_cast<const uint32_t *>(slotb + a2);
                    live = cw != NONE32;
                    if (live) {
                        const uint32_t cx2 = F::wx(cw);
                        fcx += (mcx[u] == cx2) ? 1u : 0u;
                        a2 = ((cw >> (2u * F::CB)) << 3) | (mcx[u] < cx2 ? 0u : 4u);
                        dl = 0;
                    }
                }
                if (!__any(live)) break;
            }
            if (go) {
                f[u] |= (fcx ? 1u : 0u) | (fcy ? 2u : 0u);
                sl[u] = a2;               // the empty slot under the last top-tree node: bid for it
                sh[u] = dl ? F::CB : 0u;  // its winner splits on the other dimension
                fb[u] = dl ? 1u : 2u;
            }
        }
    }
    PX_STOP(2, sl[u] ^ (f[u] << 28) ^ sh[u]);
    __syncthreads();  // every walk is done before the first bid changes a slot
#pragma unroll
    for (int u = 0; u < PPT; u++) {
        const uint32_t i = tid + u * T;
        if (i < n && i >= KTOP) atomicMin(reinterpret_cast<uint32_t *>(slotb + sl[u]), me[u]);
    }
    __syncthreads();
    ECAL_PHASE_MARK(13);
    // B.2: level-synchronous bidding below the top tree
    for (;;) {
        bool active = false;
#pragma unroll
        for (int u = 0; u < PPT; u++)
            if (sl[u] != ~0u) active |= px_level_step<CAP>(slotb, tid + u * T, me[u], sl[u], sh[u], fb[u], f[u]);
#ifdef ECAL_PHASE_PROF
        levels__++;
#endif
        if (!block_any(active, anyf, any_round)) break;
    }
    ECAL_PHASE_MARK(14);
    ECAL_PHASE_MARK(0);
    ECAL_PHASE_COUNT(8, levels__);
    PX_STOP(3, sl[u] ^ (f[u] << 28));
    // the finished tree goes out for the member-order kernel of the exact extraction (ecal_bfs.hip), which would otherwise
    // replay these insertions for every segment it is given: child links as two u16 (0xFFFF = none) per point
    if (tree_out) {
#pragma unroll
        for (int u = 0; u < PPT; u++) {
            const uint32_t i = tid + u * T;
            if (i < n) {
                const uint32_t l = slot[2 * i], r = slot[2 * i + 1];
                tree_out[base + i] = (l == NONE32 ? 0xFFFFu : (l >> (2u * F::CB))) | ((r == NONE32 ? 0xFFFFu : (r >> (2u * F::CB))) << 16);
            }
        }
        __syncthreads();   // (the bitmap below takes the child slots' place)
    }

    // ---------------- bitmap of the points (the child slots are dead: same LDS) ----------------
    {   // 16 bytes per store (the region is 16-byte aligned and PX_WORDS + 4 words long: rounding up stays inside)
        uint4 *const bm4 = reinterpret_cast<uint4 *>(bm);
        for (uint32_t k = tid; k < (H * RW + 3u) / 4u; k += T) bm4[k] = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    bool bad = false;
#pragma unroll
    for (int u = 0; u < PPT; u++) {
        const uint32_t i = tid + u * T;
        if (i < n) {
            const uint32_t bit = 1u << (mcx[u] & 31u);
            if (atomicOr(&bm[myy[u] * RW + (mcx[u] >> 5)], bit) & bit) bad = true;  // duplicate pixel: not representable
        }
    }
    __syncthreads();
    ECAL_PHASE_MARK(1);
    // raster ranks: per-word prefix inside each row (u8: a row holds at most 255 points here), then row starts
    for (uint32_t r = tid; r < H; r += T) {
        uint32_t acc = 0;
        for (uint32_t w = 0; w < RW; w++) {
            wpre[r * RW + w] = (uint8_t) acc;
            acc += (uint32_t) __popc(bm[r * RW + w]);
        }
        if (acc > 255u) bad = true;
        rowstart[r] = (uint16_t) acc;  // row count for now
    }
    if (block_any(bad, anyf, any_round)) PX_BAIL();
    {
        const uint32_t per = (H + T - 1) / T, r0 = tid * per;
        uint32_t sum = 0;
        for (uint32_t r = r0; r < r0 + per && r < H; r++) sum += rowstart[r];
        uint32_t total;
        uint32_t run = block_exscan<T>(sum, red, &total);
        for (uint32_t r = r0; r < r0 + per && r < H; r++) {
            const uint32_t c = rowstart[r];
            rowstart[r] = (uint16_t) run;
            run += c;
        }
    }
    __syncthreads();
    auto rank_of = [&](uint32_t cx, uint32_t yy) -> uint32_t {
        const uint32_t w = yy * RW + (cx >> 5);
        return (uint32_t) rowstart[yy] + (uint32_t) wpre[w] + (uint32_t) __popc(__builtin_amdgcn_ubfe(bm[w], 0u, cx & 31u));
    };
    // the 2 Rd + 1 columns from c0 of one bitmap row (word index a = row * RW + (c0 >> 5), sh5 = c0 & 31)
    auto window = [&](uint32_t a, uint32_t sh5) -> uint32_t { return __builtin_amdgcn_alignbit(bm[a + 1], bm[a], sh5); };
    auto mask_of = [&](int k) -> uint32_t { return E2I > 0 ? px_disc_mask(E2I, px_isqrt(E2I), k) : dm[k]; };
    uint32_t myrk[PPT];
#pragma unroll
    for (int u = 0; u < PPT; u++) {
        const uint32_t i = tid + u * T;
        myrk[u] = 0;
        if (i < n) {
            myrk[u] = rank_of(mcx[u], myy[u]);
            pf[myrk[u]] = (uint16_t) (i | (f[u] << F::PB));
        }
    }
    __syncthreads();
    ECAL_PHASE_MARK(5);
    PX_STOP(4, myrk[u] + pf[i]);
    // ---------------- D: core test ----------------
    bool core[PPT];
    uint32_t hdw[PPT];  // compiled-in disc: the half disc of the point as one packed word, taken from the windows phase D reads anyway
#pragma unroll
    for (int u = 0; u < PPT; u++) {
        const uint32_t i = tid + u * T;
        core[u] = false;
        hdw[u] = 0;
        if (wbase + u * T < n && i < n) {
            const uint32_t cx = mcx[u], yy = myy[u], c0 = cx - (uint32_t) Rd, sh5 = c0 & 31u;
            uint32_t a = (yy - (uint32_t) Rd) * RW + (c0 >> 5);
            uint32_t cnt = 0, vmid = 0, vlast = 0;
            if constexpr (E2I > 0) {
#pragma unroll
                for (int k = 0; k <= 2 * px_isqrt(E2I); k++) {
                    const uint32_t v = window(a, sh5);
                    cnt += (uint32_t) __popc(v & mask_of(k));
                    if (k == px_isqrt(E2I)) vmid = v;
                    if (k == 2 * px_isqrt(E2I)) vlast = v;
                    if (k <= px_isqrt(E2I) && px_hd_pos(E2I, px_isqrt(E2I), px_isqrt(E2I) + 1) <= 32)
                        hdw[u] |= __builtin_amdgcn_ubfe(v, (uint32_t) px_hd_low(E2I, px_isqrt(E2I), k), (uint32_t) px_hd_width(E2I, px_isqrt(E2I), k))
                                  << px_hd_pos(E2I, px_isqrt(E2I), k);
                    a += RW;
                }
            } else {
                for (int k = 0; k <= 2 * Rd; k++) {
                    const uint32_t v = window(a, sh5);
                    cnt += (uint32_t) __popc(v & mask_of(k));
                    vmid = (k == Rd) ? v : vmid;
                    vlast = v;
                    a += RW;
                }
            }
            cnt -= 1u;  // the point itself
            if (eps_int && cnt >= minpts) {
                // a neighbour at exactly (+eps, 0) / (0, +eps) carrying the matching bit is invisible from here
                if ((vmid >> (2 * Rd)) & 1u)
                    if (pf[rank_of(cx + (uint32_t) Rd, yy)] & (1u << F::PB)) cnt--;
                if ((vlast >> Rd) & 1u)
                    if (pf[rank_of(cx, yy + (uint32_t) Rd)] & (2u << F::PB)) cnt--;
            }
            core[u] = cnt >= minpts;
            parent[i] = core[u] ? i : NONE32;
            // the core bit joins the table entry right away: concurrent readers of this phase only look at the f
            // bits, which are the same in the old and the new value
            if (core[u]) pf[myrk[u]] = (uint16_t) (i | (f[u] << F::PB) | F::PF_CORE);
        }
    }
    __syncthreads();
    ECAL_PHASE_MARK(2);
    PX_STOP(5, parent[i] + pf[i]);
    // ---------------- E.1: union-find over the half disc (rows above, own row to the left) ----------------
    // d2 <= tlim  <=>  two core points are joined by a two-way edge (in the ball, and not an exactly-eps pair that the
    // quirk may have cut one way)
    const int tlim = eps_int ? e2i - 1 : e2i;
#pragma unroll
    for (int u = 0; u < PPT; u++) {
        const uint32_t i = tid + u * T;
        if (!(wbase + u * T < n) || !core[u]) continue;
        const uint32_t cx = mcx[u], yy = myy[u], c0 = cx - (uint32_t) Rd, sh5 = c0 & 31u;
        const uint32_t a0 = (yy - (uint32_t) Rd) * RW + (c0 >> 5);
        const uint32_t fi = f[u];
        uint32_t ri = i;  // current root of i's component
        // Anchors = neighbours already joined with i (as k, b).  A neighbour j within a two-way edge of an anchor a
        // needs no union and not even a look at its table entry: if j is core, a and j are joined by the scan of
        // whichever of the two comes later in raster order (induction on that order), and a is joined with i.
        // (Compiled-in disc: the positions an anchor settles are a precomputed mask, hd_cover, cleared from the packed word.)
        uint32_t ab0 = 4096, ak0 = 0, ab1 = 4096, ak1 = 0;
        // returns true when neighbour (k, b) is a core point now joined with i by a two-way edge
        auto link = [&](uint32_t k, uint32_t b) -> bool {
            if constexpr (E2I == 0) {   // run-time disc: the anchors are tested here; the compiled-in disc clears them from its word
                const int ex0 = (int) b - (int) ab0, ey0 = (int) k - (int) ak0, ex1 = (int) b - (int) ab1, ey1 = (int) k - (int) ak1;
                if (__mul24(ex0, ex0) + __mul24(ey0, ey0) <= tlim || __mul24(ex1, ex1) + __mul24(ey1, ey1) <= tlim) return false;
            }
            const uint32_t nx = c0 + b, ny = yy - (uint32_t) Rd + k;
            const uint32_t pfj = pf[rank_of(nx, ny)];
            if (!(pfj & F::PF_CORE)) return false;
            const uint32_t pj = pfj & F::PF_PID;
            // j = i - eps e_d: the query from j misses i exactly when i carries bit d; the query from i always
            // finds j (pruning only hides neighbours on the + side) -> one-way edge i -> j
            if (eps_int && fi) {
                const bool one_way = (k == (uint32_t) Rd && b == 0u && (fi & 1u)) || (k == 0u && b == (uint32_t) Rd && (fi & 2u));
                if (one_way) {
                    const uint32_t at = atomicAdd(n_edges, 1u);
                    if (at < PX_EDGE_CAP) {
                        edges[2 * at] = i;
                        edges[2 * at + 1] = pj;
                    }
                    return false;
                }
            }
            uint32_t rj = uf_find<false>(parent, pj);
            for (;;) {
                ri = uf_find<false>(parent, ri);
                if (ri == rj) break;
                const uint32_t hi = max(ri, rj), lo = min(ri, rj);
                if (atomicCAS(&parent[hi], hi, lo) == hi) {
                    ri = lo;
                    break;
                }
                rj = uf_find<false>(parent, rj);
            }
            ab1 = ab0;
            ak1 = ak0;
            ab0 = b;
            ak0 = k;
            return true;
        };
        if constexpr (E2I > 0) {
            // the whole half disc as one packed word: a field per row, one loop over its set bits
            constexpr int RD = px_isqrt(E2I);
            static_assert(RD <= 4 && px_hd_pos(E2I, RD, RD + 1) <= 32, "the packed half disc must fit 32 bits");
            uint32_t nm = hdw[u];  // (the bitmap has not changed since phase D)
            while (nm) {
                const uint32_t p = (uint32_t) __ffs((int) nm) - 1u;
                nm &= nm - 1u;
                const uint32_t e = hd[p];
                if (link(e >> 5, e & 31u)) nm &= ~dm[p];   // joined: everything within a two-way edge of it is settled
            }
        } else {
            unsigned long long list = 0;  // up to 8 neighbours: k << 5 | bit index, one byte each (k = dy + Rd <= 7)
            uint32_t nlist = 0;
            bool overflow = Rd > 7;
            if (!overflow) {
                uint32_t a = a0;
                for (int k = 0; k <= Rd; k++) {
                    uint32_t m = window(a, sh5) & mask_of(k);
                    if (k == Rd) m &= (1u << Rd) - 1u;  // own row: strictly left of the point
                    while (m) {
                        const uint32_t b = (uint32_t) __ffs((int) m) - 1u;
                        m &= m - 1u;
                        if (nlist < 8u) list |= (unsigned long long) (((uint32_t) k << 5) | b) << (8u * nlist);
                        else overflow = true;
                        nlist++;
                    }
                    a += RW;
                }
            }
            if (!overflow) {
                for (uint32_t q = 0; q < nlist; q++) {
                    const uint32_t e = (uint32_t) (list >> (8u * q)) & 0xFFu;
                    link(e >> 5, e & 31u);
                }
            } else {  // more than 8 earlier neighbours (or a wide disc): walk the windows again
                uint32_t a = a0;
                for (int k = 0; k <= Rd; k++) {
                    uint32_t m = window(a, sh5) & mask_of(k);
                    if (k == Rd) m &= (1u << Rd) - 1u;
                    while (m) {
                        const uint32_t b = (uint32_t) __ffs((int) m) - 1u;
                        m &= m - 1u;
                        link((uint32_t) k, b);
                    }
                    a += RW;
                }
            }
        }
    }
    __syncthreads();
    ECAL_PHASE_MARK(6);
    PX_STOP(6, parent[i] + *n_edges);
    const uint32_t m_edges = *n_edges;
    if (m_edges > PX_EDGE_CAP) PX_BAIL();
    // ---------------- E.2 flatten; E.3 one-way edges to the fix-point ----------------
    uint32_t root[PPT];  // label of the point = smallest pid of its cluster (NONE32: not a core point)
#pragma unroll
    for (int u = 0; u < PPT; u++) {
        root[u] = NONE32;
        if (core[u]) {
            root[u] = uf_root<false>(parent, tid + u * T);  // read-only walk: see uf_root
            parent[tid + u * T] = root[u];
        }
    }
    if (m_edges > 0) {
        __syncthreads();
        uint32_t *const comp = bm;  // the bitmap is dead
        for (uint32_t i = tid; i < n; i += T) comp[i] = i;
        __syncthreads();
        for (;;) {
            bool changed = false;
            for (uint32_t e = tid; e < m_edges; e += T) {
                const uint32_t ru = parent[edges[2 * e]], rv = parent[edges[2 * e + 1]];
                const uint32_t lu = comp[ru];
                if (lu < comp[rv]) {
                    atomicMin(&comp[rv], lu);
                    changed = true;
                }
            }
            if (!block_any(changed, anyf, any_round)) break;
        }
#pragma unroll
        for (int u = 0; u < PPT; u++)
            if (core[u]) root[u] = comp[root[u]];
    }
    ECAL_PHASE_MARK(3);
    ECAL_PHASE_COUNT(10, 1);
    ECAL_PHASE_COUNT(11, m_edges);
    // ---------------- F: seeds ranked in pid order = reference cluster ids ----------------
    // A seed is a point that is its own label.  Batch (u, wave) holds 64 consecutive pids and the batches ascend in
    // pid: a seed's rank = seeds in the batches before + seeds on the lanes below (ballot + mbcnt; the labels are
    // still in registers, so no sweep over the label array and no block scan).
    uint16_t *const rank = pf;  // rank -> pid table is dead
    uint32_t *const bcount = red;  // [PPT * T / 64] seeds per batch
    uint32_t below[PPT];
    bool seed[PPT];
    const uint32_t lane = tid & 63u, wave = tid >> 6;
#pragma unroll
    for (int u = 0; u < PPT; u++) {
        seed[u] = core[u] && root[u] == tid + u * T;
        const unsigned long long m = __ballot(seed[u]);
        below[u] = (uint32_t) __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) bcount[u * (T / 64) + wave] = (uint32_t) __popcll(m);
    }
    __syncthreads();  // (also: every read of pf / comp above is done before rank[] is written)
    uint32_t total = 0;
    {
        uint32_t before[PPT];
#pragma unroll
        for (int u = 0; u < PPT; u++) before[u] = 0;
#pragma unroll
        for (int q = 0; q < PPT * (T / 64); q++) {
            const uint32_t c = bcount[q];
#pragma unroll
            for (int u = 0; u < PPT; u++)
                if ((uint32_t) q < u * (T / 64) + wave) before[u] += c;
            total += c;
        }
#pragma unroll
        for (int u = 0; u < PPT; u++)
            if (seed[u]) rank[tid + u * T] = (uint16_t) (before[u] + below[u]);
    }
    __syncthreads();
    int32_t *const out = labels + base;
#pragma unroll
    for (int u = 0; u < PPT; u++) {
        const uint32_t i = tid + u * T;
        if (i < n) out[i] = (root[u] == NONE32) ? -1 : (int32_t) rank[root[u]];
    }
    if (tid == 0) {
        n_clusters[s] = total;
        if (tree_flag) tree_flag[s] = tree_epoch;   // (a segment that was bailed out of keeps an older call's number: no tree)
    }
    ECAL_PHASE_MARK(4);
#undef PX_BAIL
#undef PX_STOP
    return (int) total;
}

// first pass: workgroup b handles segment b; what it cannot take goes to todo / todo_count
template <int E2I, int CAP>
__global__ __launch_bounds__(PX_T) void dbscan_pixel_kernel(const double *__restrict__ xy,
                                                            const uint32_t *__restrict__ seg_off,
                                                            const uint32_t *__restrict__ seg_cnt, const PxGeom geom,
                                                            uint32_t minpts, int32_t *__restrict__ labels,
                                                            uint32_t *__restrict__ n_clusters,
                                                            uint32_t *__restrict__ todo,
                                                            uint32_t *__restrict__ todo_count,
                                                            const uint32_t *__restrict__ xy16, const uint32_t *__restrict__ seg_fmt,
                                                            uint32_t *__restrict__ tree_out, uint32_t *__restrict__ tree_flag,
                                                            uint32_t tree_epoch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char px_smem[];
    px_segment<E2I, CAP>(px_smem, blockIdx.x, xy, seg_off, seg_cnt, geom, minpts, labels, n_clusters, todo, todo_count, 0, 0, xy16, seg_fmt,
                         tree_out, tree_flag, tree_epoch);
}

// The LATENCY form of the two passes (ecal_ctx::latency_pass: few windows hold work — the tail of the keyframe search, where a
// pass's time is the SUM of its launches' single-workgroup latencies): workgroup b takes segment b through the pass its size asks
// for, so a segment of 769 .. CAP2 points does not wait for the first pass's launch to drain before the second one starts.  Same
// device code per segment, same results; what the second-pass code cannot take goes to todo2 (the general tiers').
template <int E2I, int CAP, int CAP2>
__global__ __launch_bounds__(PX_T) void dbscan_pixel_both_kernel(const double *__restrict__ xy, const uint32_t *__restrict__ seg_off,
                                                                 const uint32_t *__restrict__ seg_cnt, const PxGeom geom, uint32_t minpts,
                                                                 int32_t *__restrict__ labels, uint32_t *__restrict__ n_clusters,
                                                                 uint32_t *__restrict__ todo, uint32_t *__restrict__ todo_count,
                                                                 uint32_t *__restrict__ todo2, uint32_t *__restrict__ todo2_count,
                                                                 const uint32_t *__restrict__ xy16, const uint32_t *__restrict__ seg_fmt,
                                                                 uint32_t *__restrict__ tree_out, uint32_t *__restrict__ tree_flag,
                                                                 uint32_t tree_epoch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char px_smem[];
    // (a segment the first pass's code bails out of — its box of pixels larger than that pass's bitmap, more one-way edges than its
    // list holds — goes on to the second pass's code in this workgroup: the first pass's list stays empty, its launch is not made)
    int r = -1;
    if (seg_cnt[blockIdx.x] <= (uint32_t) CAP)
        r = px_segment<E2I, CAP>(px_smem, blockIdx.x, xy, seg_off, seg_cnt, geom, minpts, labels, n_clusters, nullptr, nullptr, 0, 0, xy16, seg_fmt,
                                 tree_out, tree_flag, tree_epoch);
    if (r == -1) {
        __syncthreads();
        px_segment<E2I, CAP2>(px_smem, blockIdx.x, xy, seg_off, seg_cnt, geom, minpts, labels, n_clusters, todo2, todo2_count, 0, 0, xy16,
                              seg_fmt, tree_out, tree_flag, tree_epoch);
    }
}

// second pass (CAP = PX_CAP2): the workgroups share the list of segments the first pass left over
// (in_list[0 .. *in_count)); what this pass cannot take either goes to todo / todo_count for the general tiers
template <int E2I, int CAP>
#ifndef ECAL_PX2_WG
#define ECAL_PX2_WG 5
#endif
__global__ __launch_bounds__(PX_T, ECAL_PX2_WG) void dbscan_pixel_list_kernel(const double *__restrict__ xy,
                                                                 const uint32_t *__restrict__ seg_off,
                                                                 const uint32_t *__restrict__ seg_cnt, const PxGeom geom,
                                                                 uint32_t minpts, int32_t *__restrict__ labels,
                                                                 uint32_t *__restrict__ n_clusters,
                                                                 uint32_t *__restrict__ todo,
                                                                 uint32_t *__restrict__ todo_count,
                                                                 const uint32_t *__restrict__ in_list,
                                                                 const uint32_t *__restrict__ in_count,
                                                                 const uint32_t *__restrict__ xy16, const uint32_t *__restrict__ seg_fmt,
                                                                 uint32_t *__restrict__ tree_out = nullptr, uint32_t *__restrict__ tree_flag = nullptr,
                                                                 uint32_t tree_epoch = 0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char px_smem[];
    const uint32_t count = *in_count;
    for (uint32_t k = blockIdx.x; k < count; k += gridDim.x) {
        // (the trees go out as the first pass's do: the exact extraction's second pass resolves its ties on them, extract_window.hpp)
        px_segment<E2I, CAP>(px_smem, in_list[k], xy, seg_off, seg_cnt, geom, minpts, labels, n_clusters, todo, todo_count, 0, 0, xy16, seg_fmt,
                             tree_out, tree_flag, tree_epoch);
        __syncthreads();
    }
}

}  // namespace ecal
