// The hot path's slicer as a device function: one window by one workgroup of PXH_T threads (EventFrame.cpp:10-36).
// Shared by the slicing kernels (ecal_events.hip) and the fused detection pass (ecal_fused.hip).
#pragma once
#include "ecal_ctx.hpp"
#include "slice_order.hpp"

#pragma clang fp contract(off)

namespace ecal {

constexpr int RECORD_BYTES = 25;

__device__ __forceinline__ double load_f64_unaligned(const uint8_t *p) {
    double v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

#ifdef ECAL_PHASE_PROF
// debug builds (tools/ro_phase_prof.py): shader-clock cycles between the barriers of the reference-order slicer, thread 0
static __device__ unsigned long long g_ro_cycles[16];
#define RO_MARK(i)                                                                                  \
    do {                                                                                            \
        if (REFORDER && LOGC == 11 && threadIdx.x == 0) {                                           \
            const unsigned long long now__ = __builtin_amdgcn_s_memtime();                          \
            atomicAdd(&g_ro_cycles[i], now__ - ro_t__);                                             \
            ro_t__ = now__;                                                                         \
        }                                                                                           \
    } while (0)
#else
#define RO_MARK(i)
#endif
#ifndef ECAL_RO_STOP
#define ECAL_RO_STOP 0   // debug builds: leave the reference-order block of slice_hash_window after phase k
#endif
#ifndef ECAL_SL_STOP
#define ECAL_SL_STOP 0   // debug builds: leave slice_pixel_kernel after phase k (tools/px_stop_probe.sh)
#endif

// ---------------- pixel windows, hash-table form: the kernel of the hot path ------------------------------------------
// Same semantics again for windows of <= 2047 events whose coordinates are sensor pixels, 0 <= x <= 2047, 0 <= y <= 1023
// (+0.0 only).  The counting sort + per-event scan of a hash bucket of slice_pixel_kernel spent 58 % of the kernel's
// VALU instructions in the bucket loops (trip count = the fullest bucket of the wave).  Here each polarity has an
// open-addressing table of 2048 words  pixel << 11 | event index : an event claims its pixel's slot with a CAS or lowers
// the index there with ds_min_u32 — first occurrence = smallest index, exactly what the sort delivered — and after a
// barrier looks its pixel up in the OTHER polarity's table (present: the pixel is erased, EventFrame.cpp:24-32).  Load
// factor ~0.3: 1.2 probes on average.  An event's pixel, polarity and representative stay in the registers of its thread
// from decode to output: the tables are all the LDS there is (16 KB).
// LOGC = 11: the first pass (<= 2047 events, x <= 2047, y <= 1023; 24 KB of LDS, six windows per CU);
// LOGC = 12: the second pass over the windows the first one lists (<= 4095 events, x, y <= 1023; 49 KB, three per CU).
// LOGC = 13 (round 6): the third pass, reference order only (<= 5119 events, x <= 511, y <= 1023; 66 KB, two per CU) — the
// keyframe search grows its windows to nine and ten steps (eventCameraCalib.cpp:57-95), 4500 - 5000 events on the benchmark
// stream: one window in eight went to the general tier (a 512-thread kernel with 120 KB of LDS, one workgroup per CU).
constexpr int PXH_T = 256;
#ifndef ECAL_PXH13_PER
#define ECAL_PXH13_PER 20
#endif
template <int LOGC>
struct PixHash {
    static constexpr uint32_t SLOTS = 1u << LOGC;        // per polarity
    // events per thread at most (LOGC = 13, the third pass: 20 — windows of up to 5119 events, what the keyframe search's windows
    // of nine and ten steps hold; the tables' 8192 slots give the event indices their 13 bits)
    static constexpr int PER = LOGC == 13 ? ECAL_PXH13_PER : (int) (SLOTS / PXH_T);
    static constexpr uint32_t CAP = (uint32_t) PER * PXH_T - 1u;   // events per window: indices 0 .. CAP - 1, CAP = "erased"
    static constexpr uint32_t PIXB = 32u - LOGC;         // pixel bits: x << 10 | y
    static constexpr double XMAX = (double) ((1u << (PIXB - 10u)) - 1u), YMAX = 1023.0;
    static constexpr size_t tab_off = 0;                                   // u32[2][SLOTS]; later pos u16[SLOTS] + batch counts
    static constexpr size_t red_off = tab_off + 8 * SLOTS;                 // 16 x u64 + 4 x u32 flags
    static constexpr size_t bytes = red_off + 16 * 8 + 16;
    static __device__ __forceinline__ uint32_t slot(uint32_t pix) { return (pix * 0x9E3779B1u) >> (32u - LOGC); }
    // reference element order (slice_order.hpp): what takes the tables' place once they are dead.  A wave PAIR per polarity
    // runs the epochs; thread l128 of a pair owns the keys l128 + 128 i, i < NI.
    //   W u32[2 NOFF]       per sequence position: members of the bucket first seen there, then  run start << 10 | members
    //   region u16[2 NOFF]  the runs of the buckets of three and more; before the epochs the keys' epoch-7 buckets (second
    //                       pass); at the very end pos u16[SLOTS] by event index
    //   cur u16[2 NOFF]     list position of key uid after the early epochs, final index at the end
    //   fa u32[FA_CAP]      first sequence position per bucket, the + table then the - table; before the epochs the keys'
    //                       early bucket words
    //   keep u32[16 NI]     kept keys by list position (bitmaps of both sets) and the running counts of their words
    // The - set's part of W / region / cur starts at the fixed offset NOFF = 128 NI: every run of 128 positions a scan
    // touches lies inside the set's own part (no bounds tests).
#ifndef ECAL_RO_NI
#define ECAL_RO_NI 9
#endif
#ifndef ECAL_RO_NI2
#define ECAL_RO_NI2 16
#endif
    // keys per thread of a pair: 1152 >= 1109 in the first pass; the second pass takes sets of up to 2048 keys (19 would hold
    // the 2357 of its last epoch, but with 16 the layout is 52 KB instead of 60: three windows per CU instead of two — a
    // window of 4095 events has ~1600 keys per polarity; the rare larger set goes to the general tier)
    // (the third pass: 19 = 2432 keys, every set that stays within the 2357 buckets of the eighth epoch)
    static constexpr int NI = LOGC == 11 ? ECAL_RO_NI : (LOGC == 12 ? ECAL_RO_NI2 : 19);
    static constexpr int MAX_EPOCHS = LOGC == 11 ? 7 : 8;             // bucket counts up to 1109 / 2357
    static constexpr uint32_t NOFF = 128u * NI, PSL = 2u * NOFF;
#ifndef ECAL_RO_FA
#define ECAL_RO_FA 2400
#endif
    static constexpr uint32_t FA_CAP = LOGC == 11 ? (uint32_t) ECAL_RO_FA : 4800u;    // B(+) + B(-): 1109 + 1109 / 2357 + 2357 (+ slack)
    static constexpr size_t w_off = 0;
    static constexpr size_t region_off = w_off + 4 * PSL;
    static constexpr size_t cur_off = region_off + 2 * (PSL > SLOTS ? PSL : SLOTS);
    static constexpr size_t fa_off = cur_off + 2 * PSL;
    static constexpr size_t keep_off = fa_off + 4 * FA_CAP;
    static constexpr size_t bcnt_off = keep_off + 4 * 16 * NI;        // u32[PER * 4 + 1]: batch counts of the rank scan
    static constexpr size_t ored_off = bcnt_off + 4 * (SLOTS / 64 + 4);
    static constexpr size_t obytes = ored_off + 16 * 8 + 16;
};

// std::hash<double> of the integers 0 .. 2047 (the pixel kernels' coordinates), built at compile time
struct HashIntTable {
    uint64_t v[2048];
    constexpr HashIntTable() : v{} {
        for (int i = 0; i < 2048; i++) v[i] = ref_hash_f64_bits(__builtin_bit_cast(uint64_t, (double) i));
    }
};
static __device__ const HashIntTable HASH_INT = HashIntTable();

// h % B for B < 2^13 in two exact fp64 steps (64-bit integer division is a long software sequence on the GPU): with
// inv = (1 / B)(1 - 2^-50) the estimate trunc(v inv) is the quotient or one less for every v < 2^53 B / 2^13, so one
// conditional subtraction finishes a step; h = d1 2^40 + d0, d1 < 2^24:  (d1 % B) 2^40 + d0 < 2^53.
struct ModB {
    double b, inv;
};
__host__ __device__ constexpr double ref_step_inv(int e) { return (1.0 / (double) ref_bucket_step(e)) * (1.0 - 0x1p-50); }
struct StepInvTable {
    double inv[12];
    constexpr StepInvTable() : inv{} {
        for (int e = 0; e < 12; e++) inv[e] = ref_step_inv(e);
    }
};
__device__ __forceinline__ ModB mod_for_epoch(int e) {
    constexpr StepInvTable tabl = StepInvTable();
    ModB m;
    m.b = (double) ref_bucket_step(e);
    m.inv = tabl.inv[e];
    return m;
}
__device__ __forceinline__ double mod_step(double v, const ModB m) {
    const double q = __builtin_trunc(v * m.inv);
    double r = __builtin_fma(-q, m.b, v);
    if (r >= m.b) r -= m.b;
    return r;
}
__device__ __forceinline__ uint32_t mod_hash(uint64_t h, const ModB m) {
    const double d1 = (double) (uint32_t) (h >> 40);
    const double d0 = __builtin_fma((double) (uint32_t) ((h >> 32) & 0xFFu), 0x1p32, (double) (uint32_t) h);
    const double r1 = mod_step(d1, m);
    return (uint32_t) mod_step(__builtin_fma(r1, 0x1p40, d0), m);
}

// workgroup barrier that orders LDS traffic only: global loads issued before it stay in flight (__syncthreads() drains them)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Inclusive add-scan over the 64 lanes in registers (DPP row shifts + row broadcasts, the gfx9 sequence): seven VALU
// instructions and no LDS round trip; __shfl_up costs a ds_bpermute per step.
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t x) {
    uint32_t t = x;
    t += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x111, 0xF, 0xF, false);   // row_shr:1
    t += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x112, 0xF, 0xF, false);   // row_shr:2
    t += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x113, 0xF, 0xF, false);   // row_shr:3
    t += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) t, 0x114, 0xF, 0xE, false);   // row_shr:4, banks 1-3
    t += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) t, 0x118, 0xF, 0xC, false);   // row_shr:8, banks 2-3
    t += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) t, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
    t += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) t, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
    return t;
}

// The early epochs (13, 29, 59, 127 buckets) of one polarity's set, run by ONE wave (no workgroup barrier): keys
// u < min(m, 127), at most two per lane.  In: fa[u] = the key's four bucket numbers packed 4 + 5 + 6 + 7 bits (the table
// itself takes that place afterwards).  W / region: 128-entry scratch of this polarity.  Out: cur_out[u] = list position
// after the last early epoch.
__device__ __forceinline__ void early_epochs_packed(uint32_t m, uint32_t *fa, uint32_t *W, uint16_t *region, uint16_t *cur_out) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t m_e = m < 127u ? m : 127u;
    uint32_t bw[2], cur[2] = {0u, 0u};
#pragma unroll
    for (int i = 0; i < 2; i++) bw[i] = (lane + 64u * i < m_e) ? fa[lane + 64u * i] : 0u;
    wave_sync_lds();
    uint32_t n_prev = 0;
    for (int e = 0; e < 4 && n_prev < m_e; e++) {
        const uint32_t B = (uint32_t) ref_bucket_step(e);
        const uint32_t n_e = m_e < B ? m_e : B;
        const uint32_t sh = e == 0 ? 0u : (e == 1 ? 4u : (e == 2 ? 9u : 15u)), mask = (16u << e) - 1u;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            if (lane + 64u * i < B) fa[lane + 64u * i] = 0xFFFFFFFFu;   // (the other polarity's tables may start right behind)
            if (lane + 64u * i < m_e) W[lane + 64u * i] = 0u;
        }
        wave_sync_lds();
        uint32_t b[2], q[2], f[2], sl[2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const uint32_t u = lane + 64u * i;
            q[i] = u < n_prev ? cur[i] : u;
            b[i] = (bw[i] >> sh) & mask;
            if (u < n_e) atomicMin(&fa[b[i]], q[i]);
        }
        wave_sync_lds();
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const uint32_t u = lane + 64u * i;
            f[i] = 0;
            sl[i] = 0;
            if (u < n_e) {
                f[i] = fa[b[i]];
                sl[i] = atomicAdd(&W[f[i]], 1u);
            }
        }
        wave_sync_lds();
        {   // exclusive scan of the counts over the positions 2 lane, 2 lane + 1
            const uint32_t c0 = 2u * lane < n_e ? W[2u * lane] : 0u, c1 = 2u * lane + 1u < n_e ? W[2u * lane + 1u] : 0u;
            const uint32_t ex = wave_incl_scan_dpp(c0 + c1) - c0 - c1;
            if (2u * lane < n_e) W[2u * lane] = (ex << 16) | c0;
            if (2u * lane + 1u < n_e) W[2u * lane + 1u] = ((ex + c0) << 16) | c1;
        }
        wave_sync_lds();
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const uint32_t u = lane + 64u * i;
            if (u < n_e) region[(W[f[i]] >> 16) + sl[i]] = (uint16_t) q[i];
        }
        wave_sync_lds();
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const uint32_t u = lane + 64u * i;
            if (u < n_e) {
                const uint32_t w = W[f[i]], b0 = w >> 16, c = w & 0xFFFFu;
                uint32_t within = 0;
                for (uint32_t t = 0; t < c; t++) within += ((uint32_t) region[b0 + t] < q[i]) ? 1u : 0u;
                cur[i] = n_e - 1u - (b0 + within);
            }
        }
        wave_sync_lds();
        n_prev = n_e;
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
        if (lane + 64u * i < m_e) cur_out[lane + 64u * i] = (uint16_t) cur[i];
}

// Returns false when the window is not this pass's (too many events or keys, a non-pixel coordinate): it is then put on `todo`, or —
// todo == nullptr — left to the caller (nothing has been written for it yet except, with seg_fmt, its "doubles" mark).
template <int LOGC, bool REFORDER>
__device__ __forceinline__ bool slice_hash_window(unsigned char *smem, const uint32_t s, const uint8_t *__restrict__ rec,
                                                  const uint32_t *__restrict__ win_lo, const uint32_t *__restrict__ win_hi,
                                                  const uint32_t *__restrict__ win_base, uint32_t cap_points,
                                                  double *__restrict__ xy_out, uint32_t *__restrict__ seg_off,
                                                  uint32_t *__restrict__ seg_cnt, int32_t *__restrict__ event_point,
                                                  int *overflow, uint32_t *__restrict__ todo,
                                                  uint32_t *__restrict__ todo_count, const uint2 *__restrict__ bucket_tab = nullptr,
                                                  uint32_t *__restrict__ xy16 = nullptr, uint32_t *__restrict__ seg_fmt = nullptr) {
    // xy16 / seg_fmt (ecal_packed_points): the window's points go out as x | y << 16 (4 bytes a point instead of 16: these ARE
    // sensor pixels) and its two segments are marked 1 = "packed only"; the doubles are made on request (ecal_unpack_points_dev)
    using L = PixHash<LOGC>;
    constexpr int T = PXH_T;
#ifdef ECAL_PHASE_PROF
    unsigned long long ro_t__ = __builtin_amdgcn_s_memtime();
#endif
    constexpr uint32_t PXH_CAP = L::CAP, PXH_SLOTS = L::SLOTS, IDXM = L::SLOTS - 1u, PIXB = L::PIXB;
    // the per-event word of the reference-order block: first occurrence (MB bits) | erased | polarity | key | rank << 16
    constexpr uint32_t MB = LOGC <= 12 ? 12u : 13u, M_IDX = (1u << MB) - 1u, M_ER = 1u << MB, M_POL = 2u << MB, M_KEY = 4u << MB;
    static_assert(MB + 3u <= 16u && L::CAP <= M_IDX, "the word's fields");
    constexpr int PXH_PER = L::PER;
    constexpr uint32_t NONE = L::CAP, EMPTY = 0xFFFFFFFFu;
    const uint32_t tid = threadIdx.x;
    const uint32_t lo = win_lo[s], n = win_hi[s] - lo, base = win_base[s];
    if (n == 0) {
        if (tid == 0) {
            seg_off[2 * s] = base < cap_points ? base : 0;
            seg_off[2 * s + 1] = seg_off[2 * s];
            seg_cnt[2 * s] = 0;
            seg_cnt[2 * s + 1] = 0;
            if (seg_fmt) seg_fmt[2 * s] = seg_fmt[2 * s + 1] = 0u;
        }
        return true;
    }
    if (n > PXH_CAP) {
        if (tid == 0) {
            if (todo) todo[atomicAdd(todo_count, 1u)] = s;   // (null: the caller takes the window on itself — false comes back)
            if (seg_fmt) seg_fmt[2 * s] = seg_fmt[2 * s + 1] = 0u;   // (whoever takes the window later writes doubles)
        }
        return false;
    }
    if ((uint64_t) base + n > cap_points) {  // caller's buffers too small: report, emit empty segments
        if (tid == 0) {
            *overflow = 1;
            seg_off[2 * s] = seg_off[2 * s + 1] = 0;
            seg_cnt[2 * s] = seg_cnt[2 * s + 1] = 0;
            if (seg_fmt) seg_fmt[2 * s] = seg_fmt[2 * s + 1] = 0u;
        }
        return true;
    }
    uint32_t *const tab = reinterpret_cast<uint32_t *>(smem + L::tab_off);  // [0 .. 2047] negative, [2048 .. 4095] positive
    uint16_t *const pos = reinterpret_cast<uint16_t *>(smem + L::tab_off);
    uint32_t *const red = reinterpret_cast<uint32_t *>(smem + L::red_off);
    uint32_t *const badf = red + 32;

    // a. decode (Event.hpp:41-47); the records' loads are all issued before the tables are cleared behind them
    double vx[PXH_PER], vy[PXH_PER];
    uint32_t vp[PXH_PER];
#pragma unroll
    for (int j = 0; j < PXH_PER; j++) {
        const uint32_t k = tid + j * T;
        vx[j] = 0;
        vy[j] = 0;
        vp[j] = 0;
        if (k < n) {
            const uint8_t *r = rec + (uint64_t) (lo + k) * RECORD_BYTES;
            vx[j] = load_f64_unaligned(r + 8);
            vy[j] = load_f64_unaligned(r + 16);
            vp[j] = r[24];
        }
    }
    {
        uint4 *t4 = reinterpret_cast<uint4 *>(tab);
        for (uint32_t q = tid; q < 2 * PXH_SLOTS / 4; q += T) t4[q] = make_uint4(EMPTY, EMPTY, EMPTY, EMPTY);
    }
    bool bad = false;
    uint32_t pix[PXH_PER];
#pragma unroll
    for (int j = 0; j < PXH_PER; j++) {
        const uint32_t k = tid + j * T;
        const double x = vx[j], y = vy[j];
        // the sign bit rejects negative coordinates and -0.0 (a valid pixel for operator==, but the emitted element keeps
        // its sign: general path); the upper bounds are what the table word holds
        const bool okc = x == floor(x) && y == floor(y) && x <= L::XMAX && y <= L::YMAX && __double_as_longlong(x) >= 0 &&
                         __double_as_longlong(y) >= 0;
        bad = bad || (k < n && !okc);
        pix[j] = (((uint32_t) (int) x << 10) | ((uint32_t) (int) y & 0x3FFu)) & ((1u << PIXB) - 1u);
    }
    const bool wave_bad = __any(bad);
    if ((tid & 63) == 0) badf[tid >> 6] = wave_bad ? 1u : 0u;   // one flag word per wave
    __syncthreads();
    if (badf[0] | badf[1] | badf[2] | badf[3]) {
        if (tid == 0) {
            if (todo) todo[atomicAdd(todo_count, 1u)] = s;   // (null: the caller takes the window on itself — false comes back)
            if (seg_fmt) seg_fmt[2 * s] = seg_fmt[2 * s + 1] = 0u;   // (whoever takes the window later writes doubles)
        }
        return false;
    }
    if (ECAL_SL_STOP == 1) return true;
    // (reference order, first pass) the bucket numbers of every event's pixel are asked for NOW: the gather's latency hides
    // behind the table phases, whose barriers therefore order LDS traffic only
    constexpr bool EARLY_GATHER = REFORDER;
    uint2 bw[EARLY_GATHER ? PXH_PER : 1];
    if constexpr (EARLY_GATHER) {
#pragma unroll
        for (int j = 0; j < PXH_PER; j++) bw[j] = (tid + j * T < n) ? bucket_tab[pix[j]] : make_uint2(0u, 0u);
    }
    // b. every event into its polarity's table: the slot of its pixel ends up holding the smallest event index
    // (The first probe of all of a thread's events is read before any is looked at: the probes of different events are
    // independent, a loop per event would pay one LDS latency after the other; only the ~15 % of events whose first slot
    // holds another pixel go on probing.)
    uint32_t hs[PXH_PER];
    {
        uint32_t w0[PXH_PER];
#pragma unroll
        for (int j = 0; j < PXH_PER; j++) {
            hs[j] = L::slot(pix[j]);
            w0[j] = tab[(vp[j] ? PXH_SLOTS : 0u) + hs[j]];
        }
#pragma unroll
        for (int j = 0; j < PXH_PER; j++) {
            const uint32_t k = tid + j * T;
            if (k < n) {
                uint32_t *const t = tab + (vp[j] ? PXH_SLOTS : 0u);
                const uint32_t mine = (pix[j] << LOGC) | k;
                uint32_t h = hs[j], w = w0[j];
                for (;;) {
                    if (w == EMPTY) w = atomicCAS(&t[h], EMPTY, mine);   // EMPTY back: the slot is mine
                    if (w == EMPTY) break;
                    if ((w >> LOGC) == pix[j]) {
                        atomicMin(&t[h], mine);
                        break;
                    }
                    h = (h + 1u) & (PXH_SLOTS - 1u);
                    w = t[h];
                }
                hs[j] = h;
            }
        }
    }
    if constexpr (EARLY_GATHER) lds_barrier(); else __syncthreads();
    if (ECAL_SL_STOP == 2) return true;
    // c. representative = first occurrence of the pixel with this polarity, unless the pixel also fired with the other one
    uint32_t repk[PXH_PER];  // representative of event tid + j T (NONE: erased)
    uint32_t firstk[PXH_PER];  // (reference order) its set's key = first occurrence of the pixel with the polarity | erased << 15
    {
        uint32_t fw[PXH_PER], ow[PXH_PER];
#pragma unroll
        for (int j = 0; j < PXH_PER; j++) {   // own slot and first probe of the other table: all reads first
            fw[j] = tab[(vp[j] ? PXH_SLOTS : 0u) + hs[j]];
            ow[j] = tab[(vp[j] ? 0u : PXH_SLOTS) + L::slot(pix[j])];
        }
#pragma unroll
        for (int j = 0; j < PXH_PER; j++) {
            const uint32_t k = tid + j * T;
            if (k < n) {
                const uint32_t first = fw[j] & IDXM;
                const uint32_t *const o = tab + (vp[j] ? 0u : PXH_SLOTS);
                uint32_t h = L::slot(pix[j]), w = ow[j];
                bool both = false;
                for (;;) {
                    if (w == EMPTY) break;
                    if ((w >> LOGC) == pix[j]) {
                        both = true;
                        break;
                    }
                    h = (h + 1u) & (PXH_SLOTS - 1u);
                    w = o[h];
                }
                repk[j] = both ? NONE : first;
                firstk[j] = first | (both ? 0x8000u : 0u);
            } else {
                repk[j] = NONE;
                firstk[j] = 0xFFFFu;
            }
        }
    }
    if constexpr (EARLY_GATHER) lds_barrier(); else __syncthreads();  // the tables are dead from here: pos takes their place
    RO_MARK(1);
    if (ECAL_SL_STOP == 3) return true;
    if constexpr (REFORDER) {
        // ---- the reference's element order (slice_order.hpp; EventFrame.cpp:12-13,34-35) ----
        // Keys = first occurrences (before the cancellation).  Per event one word  meta = first | erased << 12 | polarity << 13
        // | key << 14 | rank << 16; per key its bucket numbers for every epoch it can go through, packed (the divisions were
        // done once per sensor pixel, bucket_table_kernel; the second pass's epoch with 2357 buckets is divided here).
        constexpr uint32_t N_EARLY = 127u;             // keys of the epochs 13, 29, 59, 127: one wave per polarity
        uint32_t *const W = reinterpret_cast<uint32_t *>(smem + L::w_off);
        uint16_t *const region = reinterpret_cast<uint16_t *>(smem + L::region_off);
        uint16_t *const cur = reinterpret_cast<uint16_t *>(smem + L::cur_off);
        uint32_t *const fa = reinterpret_cast<uint32_t *>(smem + L::fa_off);
        uint32_t *const keepW = reinterpret_cast<uint32_t *>(smem + L::keep_off);
        uint32_t *const ored = reinterpret_cast<uint32_t *>(smem + L::ored_off);
        uint16_t *const posE = region;
        const uint32_t lane = tid & 63u, wave = tid >> 6;
        // d'. rank of every key among its polarity's keys, in event order = the order in which the set saw them
        constexpr uint32_t NBATCH = (uint32_t) PXH_PER * (T / 64);
        uint32_t *const bcnt = reinterpret_cast<uint32_t *>(smem + L::bcnt_off);
        uint32_t meta[PXH_PER];
#pragma unroll
        for (int j = 0; j < PXH_PER; j++) {
            const uint32_t k = tid + j * T;
            const bool isu = k < n && (firstk[j] & 0x7FFFu) == k;
            const unsigned long long mP = __ballot(isu && vp[j] != 0), mN = __ballot(isu && vp[j] == 0);
            const unsigned long long lower = (1ull << lane) - 1ull;
            const uint32_t below = (uint32_t) __popcll((vp[j] ? mP : mN) & lower);
            if (lane == 0) bcnt[j * (T / 64) + wave] = (uint32_t) __popcll(mP) | ((uint32_t) __popcll(mN) << 16);
            meta[j] = k < n ? ((firstk[j] & M_IDX) | ((firstk[j] & 0x8000u) ? M_ER : 0u) | (vp[j] ? M_POL : 0u) |
                               (isu ? M_KEY : 0u) | (below << 16))
                            : M_ER;   // (no event: "erased", not a key)
        }
        if constexpr (EARLY_GATHER) lds_barrier(); else __syncthreads();
    RO_MARK(2);
        if (tid < 64u) {
            if constexpr (NBATCH <= 64u) {
                const uint32_t v = tid < NBATCH ? bcnt[tid] : 0u;
                uint32_t inc = v;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t o = __shfl_up(inc, d, 64);
                    if (lane >= (uint32_t) d) inc += o;
                }
                if (tid < NBATCH) bcnt[tid] = inc - v;
                if (tid == 63u) bcnt[NBATCH] = inc;
            } else {   // (the third pass: up to 128 batches, two a lane; the packed fields stay below 2^16: <= 5119 events)
                static_assert(NBATCH <= 128u, "two batches per lane");
                const uint32_t v0 = 2u * tid < NBATCH ? bcnt[2u * tid] : 0u, v1 = 2u * tid + 1u < NBATCH ? bcnt[2u * tid + 1u] : 0u;
                uint32_t inc = v0 + v1;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t o = __shfl_up(inc, d, 64);
                    if (lane >= (uint32_t) d) inc += o;
                }
                if (2u * tid < NBATCH) bcnt[2u * tid] = inc - v0 - v1;
                if (2u * tid + 1u < NBATCH) bcnt[2u * tid + 1u] = inc - v1;
                if (tid == 63u) bcnt[NBATCH] = inc;
            }
        }
        if constexpr (EARLY_GATHER) lds_barrier(); else __syncthreads();
    RO_MARK(3);
        const uint32_t totals = bcnt[NBATCH], mP = totals & 0xFFFFu, mN = totals >> 16;
        const int EP = mP ? ref_epochs(mP) : 0, EN = mN ? ref_epochs(mN) : 0;
        {
            const uint32_t need = (EP ? (uint32_t) ref_bucket_step(EP - 1) : 0u) + (EN ? (uint32_t) ref_bucket_step(EN - 1) : 0u);
            if (need > L::FA_CAP || EP > L::MAX_EPOCHS || EN > L::MAX_EPOCHS || mP > 128u * (uint32_t) L::NI ||
                mN > 128u * (uint32_t) L::NI) {   // more keys than the bucket tables / the pair's registers hold: next tier
                if (tid == 0) {
            if (todo) todo[atomicAdd(todo_count, 1u)] = s;   // (null: the caller takes the window on itself — false comes back)
            if (seg_fmt) seg_fmt[2 * s] = seg_fmt[2 * s + 1] = 0u;   // (whoever takes the window later writes doubles)
        }
                return false;
            }
        }
        {
            // ---- one wave PAIR per polarity: early epochs (13 .. 127 buckets) by one wave each, the rest by the pair ----
            constexpr int NI = L::NI;
            const uint32_t faN = EP ? (uint32_t) ref_bucket_step(EP - 1) : 0u;   // the - set's bucket table starts here
            {
#pragma unroll
                for (int j = 0; j < PXH_PER; j++) {
                    const uint32_t ex = bcnt[j * (T / 64) + wave];
                    meta[j] += ((meta[j] & M_POL) ? (ex & 0xFFFFu) : (ex >> 16)) << 16;
                    if (meta[j] & M_KEY) {
                        const uint32_t rank = meta[j] >> 16;
                        const bool pos_ = (meta[j] & M_POL) != 0;
                        W[(pos_ ? 0u : L::NOFF) + rank] = bw[j].x | ((meta[j] & M_ER) ? 0x80000000u : 0u);
                        if (rank < N_EARLY) fa[(pos_ ? 0u : faN) + rank] = bw[j].y;
                        if (L::MAX_EPOCHS > 7 && (pos_ ? EP : EN) > 7)   // (second pass, a set of more than 1109 keys)
                            region[(pos_ ? 0u : L::NOFF) + rank] =
                                (uint16_t) mod_hash(ref_hash_combine2(HASH_INT.v[pix[j] >> 10], HASH_INT.v[pix[j] & 0x3FFu]), mod_for_epoch(7));
                    }
                }
            }
            if (ECAL_RO_STOP == 2) return true;
            // Wave pair 0 (waves 0, 1) takes the + set, pair 1 the - set; thread l128 = 0 .. 127 of a pair owns the keys
            // u = l128 + 128 i: packed bucket numbers and list position stay in its registers from here on.
            const uint32_t pol = __builtin_amdgcn_readfirstlane(wave >> 1), sub = __builtin_amdgcn_readfirstlane(wave & 1u);
            const uint32_t l128 = lane + 64u * sub;
            const uint32_t m = pol == 0u ? mP : mN;
            const int E = pol == 0u ? EP : EN;
            const uint32_t uoff = pol == 0u ? 0u : L::NOFF;
            uint32_t *const Wp = W + uoff, *const fap = fa + (pol == 0u ? 0u : faN);
            uint16_t *const regp = region + uoff, *const curp = cur + uoff;
            constexpr uint32_t KW = 4u * NI;   // bitmap words per polarity (128 NI list positions)
            uint32_t *const kW = keepW + pol * KW, *const kPre = keepW + 2u * KW + pol * KW;   // bitmaps, their running counts
            if (tid == 0) ored[10] = 0u;
            __syncthreads();
    RO_MARK(4);
            uint32_t bk[NI], cu[NI];   // (second pass: the epoch-7 bucket rides in the upper half of cu[] until that epoch)
#pragma unroll
            for (int i = 0; i < NI; i++) {
                const uint32_t u = l128 + 128u * i;
                bk[i] = u < m ? Wp[u] : 0u;
                cu[i] = (L::MAX_EPOCHS > 7 && E > 7 && u < m) ? ((uint32_t) regp[u] << 16) : 0u;
            }
            if (l128 < KW) kW[l128] = 0u;
            __syncthreads();
    RO_MARK(5);
            if (sub == 0u && m) early_epochs_packed(m, fap, Wp, regp, curp);
            __syncthreads();
    RO_MARK(6);
            if (ECAL_RO_STOP == 3) return true;
            if (l128 < (m < 127u ? m : 127u)) cu[0] = (cu[0] & 0xFFFF0000u) | (uint32_t) curp[l128];
            const int EMAX = EP > EN ? EP : EN;
            if (E > 4) {
                for (uint32_t b = l128; b < 257u; b += 128u) fap[b] = 0xFFFFFFFFu;
            }
            __syncthreads();
    RO_MARK(7);
            for (int e = 4; e < EMAX; e++) {
                const bool on = e < E;
                const uint32_t B = (uint32_t) ref_bucket_step(e), Bprev = (uint32_t) ref_bucket_step(e - 1);
                const uint32_t n_e = on ? (m < B ? m : B) : 0u;
                const uint32_t sh = e == 4 ? 0u : (e == 5 ? 9u : (e == 6 ? 19u : 16u)),
                               bmask = e == 4 ? 0x1FFu : (e == 5 ? 0x3FFu : (e == 6 ? 0x7FFu : 0xFFFu));
                const bool e7 = e == 7;   // (its buckets sit in cu[]'s upper half)
                const uint32_t per = (n_e + 127u) >> 7;   // sequence positions per thread in the scan
                uint32_t fq[NI];                          // first position of the key's bucket
                // first sequence position per bucket  (and W, last read before the barrier that ended the previous epoch, is cleared)
                for (uint32_t t = 0; t < per; t++) Wp[l128 + 128u * t] = 0u;   // (128 per positions: the scan reads them all)
#pragma unroll
                for (int i = 0; i < NI; i++) {
                    const uint32_t u = l128 + 128u * i;
                    if (128u * i >= n_e) break;
                    if (u < n_e) atomicMin(&fap[((e7 ? cu[i] : bk[i]) >> sh) & bmask], u < Bprev ? (cu[i] & 0xFFFFu) : u);
                }
                __syncthreads();
                // members per bucket, counted at the bucket's first position; the arrival number is the member's slot
#pragma unroll
                for (int i = 0; i < NI; i++) {
                    const uint32_t u = l128 + 128u * i;
                    fq[i] = 0u;
                    if (128u * i >= n_e) break;
                    if (u < n_e) {
                        const uint32_t f = fap[((e7 ? cu[i] : bk[i]) >> sh) & bmask];
                        fq[i] = f | (atomicAdd(&Wp[f], 1u) << 12);
                    }
                }
                __syncthreads();
                {   // exclusive scan of the counts: thread l128 holds the positions [l128 per, l128 per + per); the second
                    // wave's part still lacks the first wave's total (ored[]), added by the readers
                    uint32_t c[NI], sum = 0, big = 0;
#pragma unroll
                    for (int t = 0; t < NI; t++) {
                        c[t] = 0u;
                        if ((uint32_t) t < per) c[t] = Wp[l128 * per + t];
                        sum += c[t];
                        big |= c[t];
                    }
                    if (__any(big > 1023u)) ored[10] = 1u;   // a bucket of > 1023 keys does not fit the packed words: next tier
                    const uint32_t inc = wave_incl_scan_dpp(sum);
                    if (lane == 63u) ored[wave] = inc;
                    uint32_t ex = inc - sum;
#pragma unroll
                    for (int t = 0; t < NI; t++) {
                        if ((uint32_t) t < per) Wp[l128 * per + t] = (ex << 10) | (c[t] & 0x3FFu);
                        ex += c[t];
                    }
                    if (e + 1 < E) {   // the bucket table is dead (fq[] holds what was read from it): set it up for the next epoch
                        const uint32_t Bn = (uint32_t) ref_bucket_step(e + 1);
                        for (uint32_t b = l128; b < Bn; b += 128u) fap[b] = 0xFFFFFFFFu;
                    }
                }
                __syncthreads();
                const uint32_t carry_from = 64u * per, carry = __builtin_amdgcn_readfirstlane(ored[pol * 2u]);
                // buckets of three and more: the members take the slots of the bucket's run in arrival order ...
#pragma unroll
                for (int i = 0; i < NI; i++) {
                    const uint32_t u = l128 + 128u * i;
                    if (128u * i >= n_e) break;
                    if (u < n_e) {
                        const uint32_t f = fq[i] & 0xFFFu, w = Wp[f];
                        if ((w & 0x3FFu) > 2u)
                            regp[(w >> 10) + (f >= carry_from ? carry : 0u) + (fq[i] >> 12)] = (uint16_t) (u < Bprev ? (cu[i] & 0xFFFFu) : u);
                    }
                }
                __syncthreads();
                // ... and rank themselves by sequence position: new list position = n - 1 - (run start + members before it).
                // (Alone: none before it.  A bucket of two: the one whose position IS the bucket's first position is first.)
#pragma unroll
                for (int i = 0; i < NI; i++) {
                    const uint32_t u = l128 + 128u * i;
                    if (128u * i >= n_e) break;
                    if (u < n_e) {
                        const uint32_t f = fq[i] & 0xFFFu, w = Wp[f], b0 = (w >> 10) + (f >= carry_from ? carry : 0u), cnt = w & 0x3FFu;
                        const uint32_t q = u < Bprev ? (cu[i] & 0xFFFFu) : u;
                        uint32_t within = (q != f) ? 1u : 0u;
                        if (cnt > 2u) {   // (buckets of five and more are rare: four slots read at once, a loop for the rest)
                            const uint32_t r0 = regp[b0], r1 = regp[b0 + 1u], r2 = regp[b0 + 2u], r3 = regp[b0 + 3u];
                            within = (r0 < q ? 1u : 0u) + (r1 < q ? 1u : 0u) + (r2 < q ? 1u : 0u) + ((cnt > 3u && r3 < q) ? 1u : 0u);
                            for (uint32_t t = 4; t < cnt; t++) within += ((uint32_t) regp[b0 + t] < q) ? 1u : 0u;
                        }
                        cu[i] = (cu[i] & 0xFFFF0000u) | (n_e - 1u - (b0 + within));
                    }
                }
                __syncthreads();
                RO_MARK(8 + (e - 4));
            }
            if (ECAL_RO_STOP == 4) return true;
            if (ored[10]) {
                if (tid == 0) {
            if (todo) todo[atomicAdd(todo_count, 1u)] = s;   // (null: the caller takes the window on itself — false comes back)
            if (seg_fmt) seg_fmt[2 * s] = seg_fmt[2 * s + 1] = 0u;   // (whoever takes the window later writes doubles)
        }
                return false;
            }
            // the erased keys drop out (EventFrame.cpp:24-32): index of a kept key = kept keys in front of it in the list
#pragma unroll
            for (int i = 0; i < NI; i++) {
                const uint32_t u = l128 + 128u * i;
                if (u < m && !(bk[i] >> 31)) atomicOr(&kW[(cu[i] & 0xFFFFu) >> 5], 1u << (cu[i] & 31u));
            }
            __syncthreads();
    RO_MARK(11);
            if (sub == 0u) {
                static_assert(KW <= 128u, "two bitmap words per lane");
                const uint32_t c0 = 2u * lane < KW ? (uint32_t) __popc(kW[2u * lane]) : 0u;
                const uint32_t c1 = 2u * lane + 1u < KW ? (uint32_t) __popc(kW[2u * lane + 1u]) : 0u;
                const uint32_t inc = wave_incl_scan_dpp(c0 + c1);
                if (2u * lane < KW) kPre[2u * lane] = inc - c0 - c1;
                if (2u * lane + 1u < KW) kPre[2u * lane + 1u] = inc - c1;
                if (lane == 63u) ored[8u + pol] = inc;
            }
            __syncthreads();
    RO_MARK(12);
#pragma unroll
            for (int i = 0; i < NI; i++) {
                const uint32_t u = l128 + 128u * i;
                if (u < m && !(bk[i] >> 31))
                    curp[u] = (uint16_t) (kPre[(cu[i] & 0xFFFFu) >> 5] + (uint32_t) __popc(kW[(cu[i] & 0xFFFFu) >> 5] & ((1u << (cu[i] & 31u)) - 1u)));
            }
            __syncthreads();
    RO_MARK(13);
            const uint32_t nP = mP ? __builtin_amdgcn_readfirstlane(ored[8]) : 0u, nN = mN ? __builtin_amdgcn_readfirstlane(ored[9]) : 0u;
            // the event -> point map is the one output that needs every event to learn its REPRESENTATIVE's index (a table
            // by event index + a barrier); a caller that does not ask for it (event_point == nullptr) gets the points alone:
            // a key's index is its own list position
            const bool want_ep = event_point != nullptr;
            if (want_ep) {
#pragma unroll
                for (int j = 0; j < PXH_PER; j++) {
                    if ((meta[j] & (M_KEY | M_ER)) == M_KEY)   // a key, and not erased
                        posE[meta[j] & M_IDX] = cur[((meta[j] & M_POL) ? 0u : L::NOFF) + (meta[j] >> 16)];
                }
                __syncthreads();
            }
    RO_MARK(14);
            double2 *out2 = reinterpret_cast<double2 *>(xy_out) + base;
            int32_t *ep = want_ep ? event_point + base : nullptr;
#pragma unroll
            for (int j = 0; j < PXH_PER; j++) {
                const uint32_t k = tid + j * T;
                if (k < n) {
                    if (meta[j] & M_ER) {
                        if (want_ep) ep[k] = -1;
                    } else if (want_ep || (meta[j] & M_KEY)) {
                        const uint32_t at = want_ep ? posE[meta[j] & M_IDX]
                                                    : (uint32_t) cur[((meta[j] & M_POL) ? 0u : L::NOFF) + (meta[j] >> 16)];
                        if (want_ep) ep[k] = (int32_t) at;
                        if (meta[j] & M_KEY) {
                            const uint32_t slot = (meta[j] & M_POL) ? at : nP + at;
                            if (xy16) {
                                xy16[base + slot] = (pix[j] >> 10) | ((pix[j] & 0x3FFu) << 16);
                            } else {
                                double2 v;
                                v.x = (double) (pix[j] >> 10);
                                v.y = (double) (pix[j] & 0x3FFu);
                                out2[slot] = v;
                            }
                        }
                    }
                }
            }
            if (tid == 0) {
                seg_off[2 * s] = base;
                seg_cnt[2 * s] = nP;
                seg_off[2 * s + 1] = base + nP;
                seg_cnt[2 * s + 1] = nN;
                if (seg_fmt) seg_fmt[2 * s] = seg_fmt[2 * s + 1] = xy16 ? 1u : 0u;
            }
            RO_MARK(15);
            RO_MARK(0);   // (adds ~0: counts the workgroups through the number of marks... see tools/ro_phase_prof.py)
            return true;
        }
    }
    // d. ranks of the representatives in event order.  Batch (j, wave) holds 64 consecutive events and the batches ascend in
    // event index: rank = representatives of the same polarity in the batches before + on the lanes below (ballots; the
    // per-batch counts, both polarities packed in one word, are scanned by wave 0).
    constexpr uint32_t NBATCH = (uint32_t) PXH_PER * (T / 64);
    static_assert(REFORDER || NBATCH <= 64u, "the canonical-order form scans one batch per lane (first and second pass only)");
    uint32_t *const bcnt = tab + PXH_SLOTS;  // [NBATCH] counts, then exclusive prefixes; [NBATCH]: totals (the tables are dead)
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    uint32_t below[PXH_PER];
#pragma unroll
    for (int j = 0; j < PXH_PER; j++) {
        const uint32_t k = tid + j * T;
        const bool isrep = k < n && repk[j] == k;
        const unsigned long long mP = __ballot(isrep && vp[j] != 0), mN = __ballot(isrep && vp[j] == 0);
        const unsigned long long lower = (1ull << lane) - 1ull;
        below[j] = (uint32_t) __popcll((vp[j] ? mP : mN) & lower);
        if (lane == 0) bcnt[j * (T / 64) + wave] = (uint32_t) __popcll(mP) | ((uint32_t) __popcll(mN) << 16);
    }
    __syncthreads();
    if (tid < 64u) {
        const uint32_t v = tid < NBATCH ? bcnt[tid] : 0u;   // NBATCH <= 64; the packed fields stay below 2^16 (<= 4095 events)
        uint32_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(inc, d, 64);
            if (lane >= (uint32_t) d) inc += o;
        }
        if (tid < NBATCH) bcnt[tid] = inc - v;
        if (tid == 63u) bcnt[NBATCH] = inc;
    }
    __syncthreads();
    const uint32_t totals = bcnt[NBATCH], nP = totals & 0xFFFFu, nN = totals >> 16;
#pragma unroll
    for (int j = 0; j < PXH_PER; j++) {
        const uint32_t k = tid + j * T;
        if (k < n && repk[j] == k) {
            const uint32_t ex = bcnt[j * (T / 64) + wave];
            pos[k] = (uint16_t) ((vp[j] ? (ex & 0xFFFFu) : (ex >> 16)) + below[j]);
        }
    }
    __syncthreads();
    if (ECAL_SL_STOP == 4) return true;
    // e. outputs: positives first, then negatives (canonical order = first occurrence)
    double2 *out2 = reinterpret_cast<double2 *>(xy_out) + base;
    int32_t *ep = event_point ? event_point + base : nullptr;   // (null: the caller does not want the event -> point map)
#pragma unroll
    for (int j = 0; j < PXH_PER; j++) {
        const uint32_t k = tid + j * T;
        if (k < n) {
            const uint32_t r = repk[j];
            if (r == NONE) {
                if (ep) ep[k] = -1;
            } else {
                const uint32_t at = pos[r];
                if (ep) ep[k] = (int32_t) at;
                if (r == k) {
                    const uint32_t slot = vp[j] ? at : nP + at;
                    if (xy16) {
                        xy16[base + slot] = (pix[j] >> 10) | ((pix[j] & 0x3FFu) << 16);
                    } else {
                        double2 v;
                        v.x = (double) (pix[j] >> 10);
                        v.y = (double) (pix[j] & 0x3FFu);
                        out2[slot] = v;
                    }
                }
            }
        }
    }
    if (tid == 0) {
        seg_off[2 * s] = base;
        seg_cnt[2 * s] = nP;
        seg_off[2 * s + 1] = base + nP;
        seg_cnt[2 * s + 1] = nN;
        if (seg_fmt) seg_fmt[2 * s] = seg_fmt[2 * s + 1] = xy16 ? 1u : 0u;
    }
    return true;
}
}  // namespace ecal
