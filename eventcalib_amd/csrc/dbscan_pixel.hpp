// DBSCAN of one segment of event PIXELS (integer coordinates) by one workgroup, in 25 KB of LDS.
//
// Same contract as dbscan_device.hpp (labels identical to dbscan/include/dbscan.h:115-265 on top of
// dbscan/src/kdtree.cpp:106-179, including the range query's pruning quirk); this is the form the hot path
// takes: segments of <= 1024 points whose coordinates are integers (|v| <= 16383), whose bounding box padded
// by the disc radius fits a 3232-word bitmap (a 346x260 sensor does) and eps < 16.  Anything else is appended
// to a to-do list that the general kernels work off.
//
// What makes it small and why that matters: the kernel is latency bound — a workgroup is a ~45 us chain of
// dependent LDS operations and barriers, and kernel time is ~ 1 / (workgroups per CU) from 1 to 6
// (tools/occupancy_probe.sh, profiles/r01_notes.md) — so throughput is proportional to the workgroups a CU can
// hold, i.e. to 160 KB / LDS per workgroup.  For integer pixels the
// closed form of the pruning quirk collapses to two bits per point,
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
#include "dbscan_device.hpp"

namespace ecal {

#ifndef ECAL_PX_T
#define ECAL_PX_T 256
#endif
constexpr int PX_T = ECAL_PX_T;
constexpr int PX_CAP = 1024;
constexpr int PX_PPT = PX_CAP / PX_T;
constexpr uint32_t PX_WORDS = 3232;    // bitmap words: 346x260 padded by 2*4 = 354x268 bits = 268 rows x 12 words = 3216
constexpr uint32_t PX_ROWS = 472;      // rowstart[PX_ROWS + 1 + 39]: 512 u16 (tail = disc half-widths)
constexpr uint32_t PX_EDGE_CAP = 128;  // one-way edges kept (more: left to the general kernel)
constexpr int PX_RMAX = 15;

struct PixelLayout {
    // region A: points (pid order) during A/B and the bit sets; then rank -> pid, flags by rank, edge list
    static constexpr size_t p_off = 0;
    static constexpr size_t pid_off = 0;                                    // u16[1024]
    static constexpr size_t sflags_off = pid_off + 2 * PX_CAP;              // u8[1024]
    static constexpr size_t edges_off = sflags_off + PX_CAP;                // u32[2 * PX_EDGE_CAP]
    static_assert(edges_off + 8 * PX_EDGE_CAP <= 4 * PX_CAP, "rank tables must fit the point region");
    // region B: kd child slots during B; then the bitmap; after E.1: component labels (E.3)
    static constexpr size_t slot_off = 4 * PX_CAP;                          // u32[2 * 1024]
    static constexpr size_t bm_off = slot_off;                              // u32[PX_WORDS]
    static_assert(4 * PX_WORDS >= 8 * PX_CAP, "child slots must fit the bitmap region");
    static constexpr size_t parent_off = bm_off + 4 * PX_WORDS + 16;        // u32[1024] (after one spare bitmap word)
    static constexpr size_t rowstart_off = parent_off + 4 * PX_CAP;         // u16[512]
    static constexpr size_t wpre_off = rowstart_off + 1024;                 // u8[PX_WORDS]
    static constexpr size_t red_off = wpre_off + PX_WORDS;                  // u32[48]
    static constexpr size_t bytes = red_off + 4 * 48;
};

// child-slot word: pid << 22 | x' << 11 | y'  (x', y' = bitmap coordinates < 2048).  ds_min_u32 orders the bids by
// pid, and the winner's coordinates come back with its id: one dependent LDS read per tree level instead of two.
__device__ __forceinline__ uint32_t px_word(uint32_t pid, uint32_t cx, uint32_t cy) { return (pid << 22) | (cx << 11) | cy; }
__device__ __forceinline__ uint32_t px_wx(uint32_t w) { return (w >> 11) & 0x7FFu; }
__device__ __forceinline__ uint32_t px_wy(uint32_t w) { return w & 0x7FFu; }

// one tree level for one unplaced point (register state); returns true while unplaced
__device__ __forceinline__ bool px_level_step(uint32_t *slot, uint32_t i, uint32_t self, uint32_t &st, uint32_t &f) {
    using R = IdxBits<uint32_t>;
    const uint32_t cw = slot[2 * (st & R::MASK) + ((st & R::SIDE) ? 1u : 0u)];
    const uint32_t child = cw >> 22;
    if (child == i) {
        st = R::PLACED;
        return false;
    }
    const uint32_t nd = (st & R::DIR) ? 0u : 1u;
    const uint32_t sv = nd ? px_wy(self) : px_wx(self), cv = nd ? px_wy(cw) : px_wx(cw);
    const uint32_t ns = sv < cv ? 0u : 1u;  // kdtree.cpp:128-131: left iff strictly smaller
    f |= (sv == cv) ? (1u << nd) : 0u;      // `child` becomes an ancestor splitting on nd at my coordinate
    atomicMin(&slot[2 * child + ns], self);
    st = child | (nd ? R::DIR : 0u) | (ns ? R::SIDE : 0u);
    return true;
}

__global__ __launch_bounds__(PX_T) void dbscan_pixel_kernel(const double *__restrict__ xy,
                                                            const uint32_t *__restrict__ seg_off,
                                                            const uint32_t *__restrict__ seg_cnt, double eps,
                                                            uint32_t minpts, int32_t *__restrict__ labels,
                                                            uint32_t *__restrict__ n_clusters,
                                                            uint32_t *__restrict__ todo,
                                                            uint32_t *__restrict__ todo_count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char px_smem[];
    using L = PixelLayout;
    using G = GeoI16;
    using R = IdxBits<uint32_t>;
    constexpr int T = PX_T, PPT = PX_PPT;
    const uint32_t s = blockIdx.x, tid = threadIdx.x;
#ifdef ECAL_PHASE_PROF
    unsigned long long phase_t__ = __builtin_readcyclecounter(), d7__ = 0;
    uint32_t levels__ = 0;
#endif
    const uint32_t n = seg_cnt[s];
    if (n == 0) {
        if (tid == 0) n_clusters[s] = 0;
        return;
    }
    if (n > (uint32_t) PX_CAP) {
        if (tid == 0) todo[atomicAdd(todo_count, 1u)] = s;
        return;
    }
    uint16_t *const pid_s = reinterpret_cast<uint16_t *>(px_smem + L::pid_off);
    uint8_t *const sflags = reinterpret_cast<uint8_t *>(px_smem + L::sflags_off);
    uint32_t *const edges = reinterpret_cast<uint32_t *>(px_smem + L::edges_off);
    uint32_t *const slot = reinterpret_cast<uint32_t *>(px_smem + L::slot_off);
    uint32_t *const bm = reinterpret_cast<uint32_t *>(px_smem + L::bm_off);
    uint32_t *const parent = reinterpret_cast<uint32_t *>(px_smem + L::parent_off);
    uint16_t *const rowstart = reinterpret_cast<uint16_t *>(px_smem + L::rowstart_off);
    uint16_t *const hw = rowstart + PX_ROWS + 1;  // disc half-width per row offset dy + R
    uint8_t *const wpre = reinterpret_cast<uint8_t *>(px_smem + L::wpre_off);
    uint32_t *const red = reinterpret_cast<uint32_t *>(px_smem + L::red_off);
    uint32_t *const n_edges = red + 36;
    uint32_t *const rootw = red + 37;
    uint32_t *const anyf = red + 40;
    int *const bbox = reinterpret_cast<int *>(red + 44);  // min x, min y, -max x, -max y
    uint32_t any_round = 0;
    const size_t base = seg_off[s];
    const double2 *src = reinterpret_cast<const double2 *>(xy) + base;
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
        if (tid == 0) todo[atomicAdd(todo_count, 1u)] = s;         \
        return;                                                    \
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
        return;                                                                           \
    }
#else
#define PX_STOP(k, expr)
#endif
    // ---------------- A: load, pack, bounding box ----------------
    if (tid < 3) anyf[tid] = 0;
    if (tid < 4) bbox[tid] = 0x7FFFFFFF;
    if (tid == 0) *n_edges = 0;
    __syncthreads();
    uint32_t pp[PPT];
    bool fits = true;
    {
        int mnx = 0x7FFFFFFF, mny = 0x7FFFFFFF, mxx = -0x7FFFFFFF, mxy = -0x7FFFFFFF;
        // all of the thread's loads are issued before the first is used (index clamped instead of a branch around the
        // load): with the load inside `if (i < n)` the compiler waited for each one in turn — up to four serial HBM
        // round trips at the head of every workgroup
        double2 vin[PPT];
#pragma unroll
        for (int u = 0; u < PPT; u++) vin[u] = src[min(tid + u * T, n - 1u)];
#pragma unroll
        for (int u = 0; u < PPT; u++) {
            const uint32_t i = tid + u * T;
            pp[u] = 0;
            if (i < n) {
                const double2 v = vin[u];
                fits = fits && G::fits(v);
                pp[u] = G::pack(v);
                slot[2 * i] = NONE32;
                slot[2 * i + 1] = NONE32;
                const int x = G::sx(pp[u]), y = G::sy(pp[u]);
                mnx = min(mnx, x);
                mny = min(mny, y);
                mxx = max(mxx, x);
                mxy = max(mxy, y);
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
            mnx = min(mnx, __shfl_xor(mnx, o, 64));
            mny = min(mny, __shfl_xor(mny, o, 64));
            mxx = max(mxx, __shfl_xor(mxx, o, 64));
            mxy = max(mxy, __shfl_xor(mxy, o, 64));
        }
        if ((tid & 63) == 0 && mnx != 0x7FFFFFFF) {
            atomicMin(&bbox[0], mnx);
            atomicMin(&bbox[1], mny);
            atomicMin(&bbox[2], -mxx);
            atomicMin(&bbox[3], -mxy);
        }
    }
    if (block_any(!fits, anyf, any_round)) PX_BAIL();
    G geo;
    geo.init(eps);
    int Rr = (int) floor(sqrt((double) geo.e2i));
    while ((long long) (Rr + 1) * (Rr + 1) <= (long long) geo.e2i) Rr++;
    while ((long long) Rr * Rr > (long long) geo.e2i) Rr--;
    const int Rd = Rr;
    if (Rd > PX_RMAX || !(eps < 1073741824.0)) PX_BAIL();
    const int ox = bbox[0] - Rd, oy = bbox[1] - Rd;
    const uint32_t W = (uint32_t) (-bbox[2] - bbox[0] + 1 + 2 * Rd), H = (uint32_t) (-bbox[3] - bbox[1] + 1 + 2 * Rd);
    // 64-bit window fetches may read the first word of the next row (or the spare word after the last row):
    // those bits are always masked off
    const uint32_t RW = (W + 31u) >> 5;
    if (H > PX_ROWS || (uint64_t) H * RW > PX_WORDS || W > 2047u) PX_BAIL();
    uint32_t mcx[PPT], myy[PPT], me[PPT];
#pragma unroll
    for (int u = 0; u < PPT; u++) {
        mcx[u] = (uint32_t) (G::sx(pp[u]) - ox);
        myy[u] = (uint32_t) (G::sy(pp[u]) - oy);
        me[u] = px_word(tid + u * T, mcx[u] & 0x7FFu, myy[u] & 0x7FFu);
    }
    if (tid == 0) *rootw = me[0];
    PX_STOP(1, me[u] + W + H);

    // ---------------- B: kd_insert replay -> prune bits ----------------
    uint32_t st[PPT], f[PPT];
#pragma unroll
    for (int u = 0; u < PPT; u++) {
        st[u] = R::PLACED;
        f[u] = 0;
    }
    // B.0: wave 0 alone replays the first KTOP insertions (wave-synchronous, no block barrier)
    if (tid < KTOP) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t x0 = px_wx(*rootw);
        const uint32_t i = tid;
        if (i > 0 && i < n) {
            const uint32_t side = (mcx[0] < x0) ? 0u : 1u;
            f[0] |= (mcx[0] == x0) ? 1u : 0u;
            atomicMin(&slot[side], me[0]);
            st[0] = side ? R::SIDE : 0u;
        }
        for (;;) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            bool act = false;
            if (!(st[0] & R::PLACED)) act = px_level_step(slot, i, me[0], st[0], f[0]);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (!__any(act)) break;
        }
    }
    __syncthreads();
    ECAL_PHASE_MARK(12);
    // B.1: every later point walks the finished top tree (reads only); the walks of a thread's points advance
    // together so their LDS round trips overlap
    {
        const uint32_t x0 = px_wx(*rootw);
        uint32_t a[PPT], d[PPT], side[PPT];
        bool go[PPT];
#pragma unroll
        for (int u = 0; u < PPT; u++) {
            const uint32_t i = tid + u * T;
            go[u] = i < n && i >= KTOP;
            a[u] = 0;
            d[u] = 0;
            side[u] = (mcx[u] < x0) ? 0u : 1u;
            if (go[u]) f[u] |= (mcx[u] == x0) ? 1u : 0u;
        }
        for (;;) {
            uint32_t cw[PPT];
            bool any = false;
#pragma unroll
            for (int u = 0; u < PPT; u++) cw[u] = slot[go[u] ? 2 * a[u] + side[u] : 0u];
#pragma unroll
            for (int u = 0; u < PPT; u++) {
                go[u] = go[u] && cw[u] != NONE32;
                any = any || go[u];
            }
            if (!any) break;
#pragma unroll
            for (int u = 0; u < PPT; u++) {
                if (go[u]) {
                    a[u] = cw[u] >> 22;
                    d[u] ^= 1u;
                    const uint32_t sv = d[u] ? myy[u] : mcx[u], cv = d[u] ? px_wy(cw[u]) : px_wx(cw[u]);
                    side[u] = sv < cv ? 0u : 1u;
                    f[u] |= (sv == cv) ? (1u << d[u]) : 0u;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < PPT; u++) {
            const uint32_t i = tid + u * T;
            if (i < n && i >= KTOP) st[u] = a[u] | (d[u] ? R::DIR : 0u) | (side[u] ? R::SIDE : 0u);
        }
    }
    PX_STOP(2, st[u] ^ (f[u] << 28));
    __syncthreads();  // every walk is done before the first bid changes a slot
#pragma unroll
    for (int u = 0; u < PPT; u++) {
        const uint32_t i = tid + u * T;
        if (i < n && i >= KTOP) atomicMin(&slot[2 * (st[u] & R::MASK) + ((st[u] & R::SIDE) ? 1u : 0u)], me[u]);
    }
    __syncthreads();
    ECAL_PHASE_MARK(13);
    // B.2: level-synchronous bidding below the top tree
    for (;;) {
        bool active = false;
#pragma unroll
        for (int u = 0; u < PPT; u++)
            if (!(st[u] & R::PLACED)) active |= px_level_step(slot, tid + u * T, me[u], st[u], f[u]);
#ifdef ECAL_PHASE_PROF
        levels__++;
#endif
        if (!block_any(active, anyf, any_round)) break;
    }
    ECAL_PHASE_MARK(14);
    PX_STOP(3, st[u] ^ (f[u] << 28));
    ECAL_PHASE_MARK(0);
    ECAL_PHASE_COUNT(8, levels__);

    // ---------------- bitmap of the points (the child slots are dead: same LDS) ----------------
    for (uint32_t k = tid; k < H * RW; k += T) bm[k] = 0;
    if (tid <= (uint32_t) (2 * Rd)) {
        const int dy = (int) tid - Rd;
        int w = (int) floor(sqrt((double) (geo.e2i - dy * dy)));
        while ((long long) (w + 1) * (w + 1) + (long long) dy * dy <= (long long) geo.e2i) w++;
        while ((long long) w * w + (long long) dy * dy > (long long) geo.e2i) w--;
        hw[tid] = (uint16_t) w;
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
        return (uint32_t) rowstart[yy] + (uint32_t) wpre[w] + (uint32_t) __popc(bm[w] & ((1u << (cx & 31u)) - 1u));
    };
    auto window = [&](uint32_t row, uint32_t lo, uint32_t nbits) -> uint32_t {  // nbits <= 31 bits from column lo
        const uint32_t w = row * RW + (lo >> 5);
        const unsigned long long two = ((unsigned long long) bm[w + 1] << 32) | bm[w];
        return (uint32_t) (two >> (lo & 31u)) & ((1u << nbits) - 1u);
    };
    uint32_t myrk[PPT];
#pragma unroll
    for (int u = 0; u < PPT; u++) {
        const uint32_t i = tid + u * T;
        myrk[u] = 0;
        if (i < n) {
            myrk[u] = rank_of(mcx[u], myy[u]);
            pid_s[myrk[u]] = (uint16_t) i;
            sflags[myrk[u]] = (uint8_t) f[u];
        }
    }
    __syncthreads();
    ECAL_PHASE_MARK(5);
    PX_STOP(4, myrk[u] + pid_s[i] + sflags[i]);
    const bool eps_int = geo.epsi <= Rd;  // |delta| == eps needs an integral eps (then epsi == R)
    // ---------------- D: core test ----------------
    bool core[PPT];
#pragma unroll
    for (int u = 0; u < PPT; u++) {
        const uint32_t i = tid + u * T;
        core[u] = false;
        if (i < n) {
            const uint32_t cx = mcx[u], yy = myy[u];
            uint32_t cnt = 0;
            for (int dy = -Rd; dy <= Rd; dy++) {
                const uint32_t w = hw[dy + Rd];
                cnt += (uint32_t) __popc(window(yy + dy, cx - w, 2u * w + 1u));
            }
            cnt -= 1u;  // the point itself
            if (eps_int && cnt >= minpts) {
                // a neighbour at exactly (+eps, 0) / (0, +eps) carrying the matching bit is invisible from here
                const uint32_t e = (uint32_t) geo.epsi;
                if (bm[yy * RW + ((cx + e) >> 5)] >> ((cx + e) & 31u) & 1u)
                    if (sflags[rank_of(cx + e, yy)] & 1u) cnt--;
                if (bm[(yy + e) * RW + (cx >> 5)] >> (cx & 31u) & 1u)
                    if (sflags[rank_of(cx, yy + e)] & 2u) cnt--;
            }
            core[u] = cnt >= minpts;
            parent[i] = core[u] ? i : NONE32;
            // the core bit joins the flag byte right away: concurrent readers of this phase only look at bits 0-1,
            // which are the same in the old and the new byte
            if (core[u]) sflags[myrk[u]] = (uint8_t) (f[u] | 16u);
        }
    }
    __syncthreads();
    ECAL_PHASE_MARK(2);
    PX_STOP(5, parent[i] + sflags[i]);
    // ---------------- E.1: union-find over the half disc (rows above, own row to the left) ----------------
#pragma unroll
    for (int u = 0; u < PPT; u++) {
        const uint32_t i = tid + u * T;
        if (!core[u]) continue;
        const uint32_t cx = mcx[u], yy = myy[u];
        const uint32_t fi = f[u];
        unsigned long long list = 0;  // up to 8 neighbours: (dy + R) << 5 | bit index, one byte each
        uint32_t nlist = 0;
        bool overflow = false;
        for (int dy = -Rd; dy <= 0; dy++) {
            const uint32_t w = hw[dy + Rd];
            uint32_t m = window(yy + dy, cx - w, dy == 0 ? w : 2u * w + 1u);  // own row: strictly left of the point
            while (m) {
                const uint32_t b = (uint32_t) __ffs((int) m) - 1u;
                m &= m - 1u;
                if (nlist < 8u) list |= (unsigned long long) ((((uint32_t) (dy + Rd)) << 5 | b) & 0xFFu) << (8u * nlist);
                else overflow = true;
                nlist++;
            }
        }
        uint32_t ri = i;  // current root of i's component
        // Chain rule: within one row of the disc, a core neighbour at most R to the right of the previous core
        // neighbour (and not at exactly eps from it, where the quirk may cut the edge) is already joined to it by
        // its own scan of its row, so joining i with the first of such a run joins i with all of it.
        int prev_dy = 1;
        uint32_t prev_x = 0;
        auto link = [&](int dy, uint32_t b) {
            const uint32_t w = hw[dy + Rd];
            const uint32_t nx = cx - w + b, ny = yy + dy;
            const uint32_t rk = rank_of(nx, ny);
            if (!(sflags[rk] & 16u)) return;
            const uint32_t pj = pid_s[rk];
            // j = i - eps e_d: the query from j misses i exactly when i carries bit d; the query from i always
            // finds j (pruning only hides neighbours on the + side) -> one-way edge i -> j
            const bool one_way = eps_int && ((dy == 0 && cx - nx == (uint32_t) geo.epsi && (fi & 1u)) ||
                                             (nx == cx && dy == -geo.epsi && (fi & 2u)));
            if (one_way) {
                const uint32_t at = atomicAdd(n_edges, 1u);
                if (at < PX_EDGE_CAP) {
                    edges[2 * at] = i;
                    edges[2 * at + 1] = pj;
                }
                return;
            }
            {
                const uint32_t dx = nx - prev_x;
                const bool chained = dy == prev_dy && dx <= (uint32_t) Rd && !(eps_int && dx == (uint32_t) geo.epsi);
                prev_dy = dy;
                prev_x = nx;
                if (chained) return;
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
        };
        if (!overflow) {
            for (uint32_t k = 0; k < nlist; k++) {
                const uint32_t e = (uint32_t) (list >> (8u * k)) & 0xFFu;
                link((int) (e >> 5) - Rd, e & 31u);
            }
        } else {  // more than 8 earlier neighbours: walk the windows again
            for (int dy = -Rd; dy <= 0; dy++) {
                const uint32_t w = hw[dy + Rd];
                uint32_t m = window(yy + dy, cx - w, dy == 0 ? w : 2u * w + 1u);
                while (m) {
                    const uint32_t b = (uint32_t) __ffs((int) m) - 1u;
                    m &= m - 1u;
                    link(dy, b);
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
#pragma unroll
    for (int u = 0; u < PPT; u++)
        if (core[u]) parent[tid + u * T] = uf_find<false>(parent, tid + u * T);
    __syncthreads();
    if (m_edges > 0) {
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
            if (core[u]) parent[tid + u * T] = comp[parent[tid + u * T]];
        __syncthreads();
    }
    ECAL_PHASE_MARK(3);
    ECAL_PHASE_COUNT(10, 1);
    ECAL_PHASE_COUNT(11, m_edges);
    // ---------------- F: seeds ranked in pid order = reference cluster ids ----------------
    const uint32_t *const label = parent;
    uint16_t *const rank = pid_s;  // rank -> pid table is dead
    uint32_t total;
    {
        const uint32_t per = (n + T - 1) / T, i0 = tid * per;
        uint32_t mine = 0;
        for (uint32_t i = i0; i < i0 + per && i < n; i++) mine += (label[i] == i) ? 1u : 0u;
        uint32_t run = block_exscan<T>(mine, red, &total);
        for (uint32_t i = i0; i < i0 + per && i < n; i++)
            if (label[i] == i) rank[i] = (uint16_t) (run++);
    }
    __syncthreads();
    int32_t *const out = labels + base;
#pragma unroll
    for (int u = 0; u < PPT; u++) {
        const uint32_t i = tid + u * T;
        if (i < n) {
            const uint32_t Lb = label[i];
            out[i] = (Lb == NONE32) ? -1 : (int32_t) rank[Lb];
        }
    }
    if (tid == 0) n_clusters[s] = total;
    ECAL_PHASE_MARK(4);
#undef PX_BAIL
#undef PX_STOP
}

}  // namespace ecal
