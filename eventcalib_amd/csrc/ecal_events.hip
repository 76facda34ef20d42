// Ingest + time-slicing kernels: packed 25-byte event records -> per-window unique pixel sets.
//
// Replaces (reference paths):
//   event/include/opengv2/event/Event.hpp:41-47           record decode (f64 t, f64 x, f64 y, u8 p)
//   event_camera_calib/test/eventCameraCalib.cpp:154-163  multimap<double,Event_loc_pol> fill
//   event/src/EventFrame.cpp:10-36                        window [t0,t1], per-polarity unique pixel
//                                                         sets, erase pixels present in both
// One workgroup owns one window: its records are decoded straight out of the packed stream,
// bucketed by a hash of the pixel, de-duplicated inside the bucket, and compacted in
// first-occurrence order (the build's canonical pid order).
#include "ecal_ctx.hpp"
#include "slice_order.hpp"
#include "slice_hash.hpp"

#pragma clang fp contract(off)

namespace ecal {


// ---------------- window bounds: lower_bound(t0) / upper_bound(t1) on the record stream -----------
// First index whose time stamp is not < t (STRICT = false: std::lower_bound) or not <= t (STRICT = true:
// std::upper_bound).  Time stamps grow roughly linearly with the index, so the search starts from the interpolated
// position, gallops to a bracket and bisects inside it: ~8 dependent loads instead of log2(n) = 26 for 50 M events.
template <bool STRICT>
__device__ __forceinline__ uint64_t bound_search(const uint8_t *__restrict__ rec, uint64_t n, double t, double t_first, double t_last) {
    auto below = [&](uint64_t i) {  // true: the answer lies beyond i
        const double v = load_f64_unaligned(rec + i * RECORD_BYTES);
        return STRICT ? (v <= t) : (v < t);
    };
    if (n == 0) return 0;
    const double f = (t - t_first) / (t_last - t_first);  // NaN (one record, or t = NaN) compares false: g = 0
    uint64_t g = f >= 1.0 ? n - 1 : (f > 0.0 ? (uint64_t) (f * (double) (n - 1)) : 0);
    if (g > n - 1) g = n - 1;
    uint64_t lo, hi;  // the answer is in [lo, hi]
    if (below(g)) {
        lo = g + 1;
        hi = n;
        for (uint64_t w = 32;; w *= 2) {
            const uint64_t p = g + w;
            if (p >= n) break;
            if (!below(p)) {
                hi = p;
                break;
            }
            lo = p + 1;
        }
    } else {
        hi = g;
        lo = 0;
        for (uint64_t w = 32;; w *= 2) {
            if (g < w) break;
            const uint64_t p = g - w;
            if (below(p)) {
                lo = p + 1;
                break;
            }
            hi = p;
        }
    }
    while (lo < hi) {
        const uint64_t m = (lo + hi) >> 1;
        if (below(m)) lo = m + 1; else hi = m;
    }
    return lo;
}

__global__ void window_bounds_kernel(const uint8_t *__restrict__ rec, uint64_t n, const double *__restrict__ t0,
                                     const double *__restrict__ t1, uint32_t S, uint32_t *__restrict__ lo_out,
                                     uint32_t *__restrict__ hi_out) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const double a0 = t0[s], a1 = t1[s];
    double t_first = 0, t_last = 0;
    if (n) {
        t_first = load_f64_unaligned(rec);
        t_last = load_f64_unaligned(rec + (n - 1) * RECORD_BYTES);
    }
    const uint64_t lo = bound_search<false>(rec, n, a0, t_first, t_last);   // EventFrame.cpp:14
    const uint64_t up = bound_search<true>(rec, n, a1, t_first, t_last);    // EventFrame.cpp:15
    lo_out[s] = (uint32_t) lo;
    hi_out[s] = (uint32_t) (up < lo ? lo : up);
}

// Bounds AND slot bases in ONE launch (round 4; it used to be the search kernel above + a one-workgroup scan: 14 + 13 us per
// pass, a dependent pair at the head of every pass): a workgroup of 256 windows searches its bounds, scans its sizes and gets
// the sum of all earlier workgroups by a decoupled look-back — workgroup ids are handed out by a ticket, so every
// predecessor is running or done; a workgroup first publishes its own total, then wave 0 reads the 64 predecessors' words
// at once (value | state << 32 | epoch << 34; epoch = the call's number, so the words of earlier calls read as "not there
// yet" and the table never has to be wiped) until it meets one that carries its inclusive prefix.
constexpr uint32_t WB_T = 256;
__global__ __launch_bounds__(WB_T) void window_bounds_base_kernel(const uint8_t *__restrict__ rec, uint64_t n, const double *__restrict__ t0,
                                                                  const double *__restrict__ t1, uint32_t S, uint32_t *__restrict__ lo_out,
                                                                  uint32_t *__restrict__ hi_out, uint32_t *__restrict__ base_out,
                                                                  unsigned long long *status, uint32_t epoch, uint32_t *ticket) {
    __shared__ uint32_t sh_bid, sh_prefix, wsum[WB_T / 64];
    if (threadIdx.x == 0) sh_bid = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t bid = sh_bid;
    const uint32_t s = bid * WB_T + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t size = 0;
    if (s < S) {
        const double a0 = t0[s], a1 = t1[s];
        double t_first = 0, t_last = 0;
        if (n) {
            t_first = load_f64_unaligned(rec);
            t_last = load_f64_unaligned(rec + (n - 1) * RECORD_BYTES);
        }
        const uint64_t lo = bound_search<false>(rec, n, a0, t_first, t_last);   // EventFrame.cpp:14
        uint64_t up = bound_search<true>(rec, n, a1, t_first, t_last);          // EventFrame.cpp:15
        if (up < lo) up = lo;
        lo_out[s] = (uint32_t) lo;
        hi_out[s] = (uint32_t) up;
        size = (uint32_t) (up - lo);
    }
    uint32_t inc = size;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t pre = 0, total = 0;
#pragma unroll
    for (int w = 0; w < (int) (WB_T / 64); w++) {
        const uint32_t x = wsum[w];
        if (w < wave) pre += x;
        total += x;
    }
    const unsigned long long tag = (unsigned long long) epoch << 34;
    if (wave == 0) {
        uint32_t before = 0;
        if (bid == 0) {
            if (lane == 0) __hip_atomic_store(status, tag | (2ull << 32) | total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0) __hip_atomic_store(status + bid, tag | (1ull << 32) | total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            int64_t top = (int64_t) bid - 1;            // the nearest predecessor not yet accounted for
            for (;;) {
                const int64_t j = top - lane;
                unsigned long long w = 0;
                if (j >= 0) w = __hip_atomic_load(status + j, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                const bool there = j < 0 || (w >> 34) == epoch;
                const uint32_t state = there && j >= 0 ? (uint32_t) (w >> 32) & 3u : 0u;
                const unsigned long long incl = __ballot(state == 2u), miss = __ballot(!there);
                // lanes below the first inclusive word (or all 64) must be there
                const int first_incl = incl ? __builtin_ctzll(incl) : 64;
                const unsigned long long need = first_incl >= 63 ? ~0ull : ((2ull << first_incl) - 1ull);
                if (miss & need) {
                    __builtin_amdgcn_s_sleep(2);
                    continue;
                }
                uint32_t v = (j >= 0 && lane <= first_incl) ? (uint32_t) w : 0u;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
                before += v;
                if (incl || top - 64 < 0) break;
                top -= 64;
            }
            if (lane == 0) __hip_atomic_store(status + bid, tag | (2ull << 32) | (before + total), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) sh_prefix = before;
    }
    __syncthreads();
    const uint32_t b0 = sh_prefix + pre + inc - size;
    if (s < S) base_out[s] = b0;
    if (s == S - 1) base_out[S] = b0 + size;
}

// exclusive scan of (hi - lo) over the windows; base[S] = total.  One block (S is small, <= ~1e6), four windows per thread
// and round in 16-byte accesses: the rounds are a dependent chain of global loads and barriers, so there should be few of
// them (eight per thread in 4-byte accesses was slower than one: 46 against 33 us at 33 334 windows).
__global__ __launch_bounds__(1024) void window_base_kernel(const uint32_t *__restrict__ lo,
                                                           const uint32_t *__restrict__ hi, uint32_t S,
                                                           uint32_t *__restrict__ base) {
    constexpr uint32_t PER = 4;
    __shared__ uint32_t red[17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool wide = ((reinterpret_cast<uintptr_t>(lo) | reinterpret_cast<uintptr_t>(hi) | reinterpret_cast<uintptr_t>(base)) & 15u) == 0;
    uint32_t carry = 0;
    for (uint32_t s0 = 0; s0 < S; s0 += 1024 * PER) {
        const uint32_t s = s0 + threadIdx.x * PER;
        uint32_t v[PER] = {0u, 0u, 0u, 0u};
        const bool whole = wide && s + PER <= S;
        if (whole) {
            const uint4 l4 = *reinterpret_cast<const uint4 *>(lo + s), h4 = *reinterpret_cast<const uint4 *>(hi + s);
            v[0] = h4.x - l4.x;
            v[1] = h4.y - l4.y;
            v[2] = h4.z - l4.z;
            v[3] = h4.w - l4.w;
        } else {
#pragma unroll
            for (uint32_t k = 0; k < PER; k++)
                if (s + k < S) v[k] = hi[s + k] - lo[s + k];
        }
        const uint32_t mine = v[0] + v[1] + v[2] + v[3];
        uint32_t inc = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        if (lane == 63) red[wave] = inc;
        __syncthreads();
        uint32_t pre = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) {
            const uint32_t x = red[w];
            if (w < wave) pre += x;
            tot += x;
        }
        const uint32_t b0 = carry + pre + inc - mine;
        if (whole) {
            *reinterpret_cast<uint4 *>(base + s) = make_uint4(b0, b0 + v[0], b0 + v[0] + v[1], b0 + v[0] + v[1] + v[2]);
        } else {
            uint32_t run = b0;
#pragma unroll
            for (uint32_t k = 0; k < PER; k++) {
                if (s + k < S) base[s + k] = run;
                run += v[k];
            }
        }
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) base[S] = carry;
}

// flag[0] = 1 if some record's timestamp is smaller than its predecessor's
__global__ void check_sorted_kernel(const uint8_t *__restrict__ rec, uint64_t n, int *flag) {
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (i >= n) return;
    if (load_f64_unaligned(rec + i * RECORD_BYTES) < load_f64_unaligned(rec + (i - 1) * RECORD_BYTES)) *flag = 1;
}

// ---------------- per-window slicing ----------------
template <typename Idx>
struct SliceWork {
    double2 *pts;    // [n] decoded (x, y)
    uint8_t *pol;    // [n] 1 = positive
    uint32_t *bend;  // [nb + 1] bucket cursors / ends
    Idx *sorted;     // [n] event ids grouped by bucket
    Idx *rep;        // [n] representative event of this event's (pixel, polarity), NONE if erased
    uint32_t *pos;   // [n] packed exclusive counts: low 16/.. see below (LDS) or two words (global)
    uint32_t *red;   // block-scan scratch (LDS)
    uint32_t bend_words = 0;   // words of bend[] when it lives in LDS (else 0)
    uint32_t pts_words = 0;    // words of pts[] when it lives in LDS (else 0)
};

__device__ __forceinline__ uint32_t pixel_hash(double x, double y) {
    // operator== semantics: -0.0 and +0.0 are the same pixel -> canonicalise before hashing
    const uint64_t xb = (uint64_t) __double_as_longlong(x + 0.0), yb = (uint64_t) __double_as_longlong(y + 0.0);
    uint32_t h = ((uint32_t) (xb >> 32) ^ (uint32_t) xb) * 0x9E3779B1u;
    h ^= (((uint32_t) (yb >> 32) ^ (uint32_t) yb) * 0x85EBCA77u) + (h >> 13);
    h ^= h >> 16;
    h *= 0xC2B2AE3Du;
    h ^= h >> 15;
    return h;
}

template <int T>
__device__ __forceinline__ void block_exscan2(uint32_t a, uint32_t b, uint32_t *red, uint32_t *ea, uint32_t *eb,
                                              uint32_t *ta, uint32_t *tb) {
    // two independent exclusive scans (64-bit packed) in one pass
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long v = ((unsigned long long) b << 32) | a, inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    unsigned long long *r64 = reinterpret_cast<unsigned long long *>(red);
    if (lane == 63) r64[wave] = inc;
    __syncthreads();
    unsigned long long pre = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < T / 64; w++) {
        const unsigned long long x = r64[w];
        if (w < wave) pre += x;
        tot += x;
    }
    __syncthreads();
    const unsigned long long ex = pre + inc - v;
    *ea = (uint32_t) ex;
    *eb = (uint32_t) (ex >> 32);
    *ta = (uint32_t) tot;
    *tb = (uint32_t) (tot >> 32);
}

template <bool GLOBAL>
__device__ __forceinline__ uint32_t ld_word(const uint32_t *p) {
    if constexpr (GLOBAL) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}

// EventFrame constructor for the records [lo, lo+n) of the stream.  ord == nullptr: the canonical element order (first
// occurrence); else the reference's (slice_order.hpp) with *ord as this workgroup's scratch (ord->h .. ord->fa and evk sized
// for n keys).
template <int T, bool GLOBAL, typename Idx>
__device__ __forceinline__ void slice_window(const SliceWork<Idx> wk, const uint8_t *__restrict__ rec, uint32_t lo,
                                             uint32_t n, uint32_t nb_log, double *__restrict__ xy_out,
                                             int32_t *__restrict__ event_point, uint32_t *n_pos_out,
                                             uint32_t *n_neg_out, const OrderScratch *ord, uint32_t *evk) {
    constexpr uint32_t ERASED = (sizeof(Idx) == 2) ? 0x8000u : 0x80000000u;   // flag on rep[]: the pixel fired with both polarities
    const uint32_t tid = threadIdx.x;
    double2 *const pts = wk.pts;
    uint8_t *const pol = wk.pol;
    uint32_t *const bend = wk.bend;
    Idx *const sorted = wk.sorted;
    Idx *const rep = wk.rep;
    uint32_t *const pos = wk.pos;
    const uint32_t nb = 1u << nb_log, mask = nb - 1u;

    // a. decode (Event.hpp:41-47)
    for (uint32_t b = tid; b <= nb; b += T) bend[b] = 0;
    for (uint32_t k = tid; k < n; k += T) {
        const uint8_t *r = rec + (uint64_t) (lo + k) * RECORD_BYTES;
        double2 p;
        p.x = load_f64_unaligned(r + 8);
        p.y = load_f64_unaligned(r + 16);
        pts[k] = p;
        pol[k] = r[24] ? 1 : 0;
    }
    __syncthreads();
    // b/c. counting sort of the events by pixel-hash bucket
    for (uint32_t k = tid; k < n; k += T) atomicAdd(&bend[pixel_hash(pts[k].x, pts[k].y) & mask], 1u);
    __syncthreads();
    {
        const uint32_t per = (nb + T - 1) / T, b0 = tid * per;
        uint32_t sum = 0;
        for (uint32_t b = b0; b < b0 + per && b < nb; b++) sum += ld_word<GLOBAL>(&bend[b]);
        uint32_t ex, dummy0, tot, dummy1;
        block_exscan2<T>(sum, 0u, wk.red, &ex, &dummy0, &tot, &dummy1);
        for (uint32_t b = b0; b < b0 + per && b < nb; b++) {
            const uint32_t c = ld_word<GLOBAL>(&bend[b]);
            bend[b] = ex;
            ex += c;
        }
    }
    __syncthreads();
    for (uint32_t k = tid; k < n; k += T) {
        const uint32_t at = atomicAdd(&bend[pixel_hash(pts[k].x, pts[k].y) & mask], 1u);
        sorted[at] = (Idx) k;
    }
    __syncthreads();
    // d. first occurrence per (pixel, polarity) = the key the set keeps; pixels that fired with both polarities are erased
    for (uint32_t k = tid; k < n; k += T) {
        const double2 p = pts[k];
        const uint32_t b = pixel_hash(p.x, p.y) & mask;
        uint32_t m = b ? ld_word<GLOBAL>(&bend[b - 1]) : 0u;
        const uint32_t e = ld_word<GLOBAL>(&bend[b]);
        uint32_t minP = 0xFFFFFFFFu, minN = 0xFFFFFFFFu;
        for (; m < e; m++) {
            const uint32_t j = sorted[m];
            const double2 q = pts[j];
            if (q.x == p.x && q.y == p.y) {  // std::equal_to<> on Vector2d (EventFrame.cpp:12-13)
                if (pol[j]) minP = min(minP, j); else minN = min(minN, j);
            }
        }
        const bool erased = (minP != 0xFFFFFFFFu) && (minN != 0xFFFFFFFFu);  // EventFrame.cpp:24-32
        rep[k] = (Idx) ((pol[k] ? minP : minN) | (erased ? ERASED : 0u));
    }
    __syncthreads();
    // e. rank of every key among the keys of its polarity, in event order (blocked ownership: contiguous k per thread):
    // canonical order — the keys that survive the cancellation; reference order — all keys (the order in which the set
    // saw them, EventFrame.cpp:14-21).
    const bool reforder = ord != nullptr;
    const uint32_t per = (n + T - 1) / T, k0 = tid * per;
    uint32_t cntP = 0, cntN = 0;
    for (uint32_t k = k0; k < k0 + per && k < n; k++) {
        const uint32_t r = rep[k];
        if ((r & ~ERASED) == k && (reforder || !(r & ERASED))) { if (pol[k]) cntP++; else cntN++; }
    }
    uint32_t exP, exN, nP, nN;
    block_exscan2<T>(cntP, cntN, wk.red, &exP, &exN, &nP, &nN);
    for (uint32_t k = k0; k < k0 + per && k < n; k++) {
        const uint32_t r = rep[k];
        if ((r & ~ERASED) == k && (reforder || !(r & ERASED))) pos[k] = pol[k] ? exP++ : exN++;
    }
    __syncthreads();
    if (reforder) {
        // e'. per polarity: the set's iteration order of its keys (slice_order.hpp), then the erased keys drop out
        const uint32_t mk[2] = {nN, nP};
        uint32_t kept_tot[2] = {0u, 0u};
        // the keys' hashes, both sets at once (the + set's at h[0 .. nP), the - set's behind them): in the LDS tiers the decoded
        // points then make room for the order pass — its epochs' arrays, 32 bytes a key, where the points' 16 bytes an event
        // were: epochs of up to CAP / 2 buckets (2357 for a window of 5000 events) stay out of global memory; the points
        // are decoded once more for the outputs
        for (uint32_t k = tid; k < n; k += T)
            if ((rep[k] & ~ERASED) == k) ord->h[(pol[k] ? 0u : nP) + pos[k]] = ref_pixel_hash(pts[k].x, pts[k].y);
        __syncthreads();
        for (int pl = 1; pl >= 0; pl--) {
            const uint32_t m = mk[pl];
            if (m) {
                OrderScratch w = *ord;
                if (pl == 0) w.h += nP;
                for (uint32_t k = tid; k < n; k += T)
                    if ((uint32_t) pol[k] == (uint32_t) pl && (rep[k] & ~ERASED) == k) evk[pos[k]] = k;
                __syncthreads();
                if constexpr (GLOBAL) reference_list_order<T>(w, m, wk.red);
                else reference_list_order<T, (sizeof(Idx) == 2 ? 10 : 0)>(w, m, wk.red, reinterpret_cast<uint32_t *>(pts), wk.pts_words / 8u);   // (LDS tiers: <= 10 keys a thread)
                for (uint32_t u = tid; u < m; u += T) w.cnt[w.cur[u]] = (rep[evk[u]] & ERASED) ? 0u : 1u;   // by list position
                __syncthreads();
                const uint32_t perq = (m + T - 1) / T, q0 = tid * perq;
                uint32_t sum = 0;
                for (uint32_t q = q0; q < q0 + perq && q < m; q++) sum += ld_word<GLOBAL>(&w.cnt[q]);
                uint32_t tot;
                uint32_t ex = block_exscan_u32<T>(sum, wk.red, &tot);
                for (uint32_t q = q0; q < q0 + perq && q < m; q++) {
                    w.bas[q] = ex;
                    ex += ld_word<GLOBAL>(&w.cnt[q]);
                }
                kept_tot[pl] = tot;
                __syncthreads();
                for (uint32_t u = tid; u < m; u += T) pos[evk[u]] = ld_word<GLOBAL>(&w.bas[w.cur[u]]);
                __syncthreads();
            }
        }
        nP = kept_tot[1];
        nN = kept_tot[0];
    }
    // f. outputs: positives first, then negatives
    double2 *out2 = reinterpret_cast<double2 *>(xy_out);
    for (uint32_t k = tid; k < n; k += T) {
        const uint32_t r = rep[k];
        if (r & ERASED) {
            if (event_point) event_point[k] = -1;
        } else {
            const uint32_t at = pos[r];
            if (event_point) event_point[k] = (int32_t) at;
            if (r == k) {
                double2 p;
                if (!GLOBAL && reforder) {   // (the LDS tiers' point buffer served the order pass)
                    const uint8_t *rr = rec + (uint64_t) (lo + k) * RECORD_BYTES;
                    p.x = load_f64_unaligned(rr + 8);
                    p.y = load_f64_unaligned(rr + 16);
                } else {
                    p = pts[k];
                }
                out2[pol[k] ? at : nP + at] = p;
            }
        }
    }
    *n_pos_out = nP;
    *n_neg_out = nN;
}

// scratch of the reference-order pass for windows of up to cap events: h u64[cap]; cur, slot, cnt, bas, region, evk
// u32[cap]; fa u32[9 cap / 4 + 16]
__host__ __device__ constexpr size_t order_scratch_bytes(size_t cap) {
    return ((8 + 6 * 4) * cap + 4 * (9 * cap / 4 + 16) + 255) & ~(size_t) 255;
}
__device__ __forceinline__ void order_scratch_carve(unsigned char *o, size_t cap, OrderScratch *ord, uint32_t **evk) {
    ord->h = reinterpret_cast<uint64_t *>(o);
    uint32_t *w = reinterpret_cast<uint32_t *>(o + 8 * cap);
    ord->cur = w;
    ord->slot = w + cap;
    ord->cnt = w + 2 * cap;
    ord->bas = w + 3 * cap;
    ord->region = w + 4 * cap;
    *evk = w + 5 * cap;
    ord->fa = w + 6 * cap;
}

template <int CAP>
struct SliceLayout {
    static constexpr size_t pts_off = 0;
    static constexpr size_t bend_off = pts_off + sizeof(double2) * CAP;
    static constexpr size_t pos_off = bend_off + sizeof(uint32_t) * (CAP + 4);
    static constexpr size_t sorted_off = pos_off + sizeof(uint32_t) * CAP;
    static constexpr size_t rep_off = sorted_off + sizeof(uint16_t) * CAP;
    static constexpr size_t pol_off = rep_off + sizeof(uint16_t) * CAP;
    static constexpr size_t red_off = pol_off + ((CAP + 15) / 16) * 16;
    static constexpr size_t bytes = red_off + 16 * sizeof(unsigned long long);
    static_assert(bend_off % 16 == 0 && pos_off % 16 == 0 && sorted_off % 16 == 0 && rep_off % 16 == 0 &&
                  pol_off % 16 == 0 && red_off % 16 == 0, "align");
};

template <int V>
struct Log2c {
    static constexpr uint32_t value = 1 + Log2c<V / 2>::value;
};
template <>
struct Log2c<1> {
    static constexpr uint32_t value = 0;
};

template <int CAP, int NB, int T>
__device__ __forceinline__ void slice_tier_window(unsigned char *smem, uint32_t s, const uint8_t *__restrict__ rec,
                                                  const uint32_t *__restrict__ win_lo,
                                                  const uint32_t *__restrict__ win_hi,
                                                  const uint32_t *__restrict__ win_base, uint32_t lo_excl,
                                                  uint32_t cap_points, double *__restrict__ xy_out,
                                                  uint32_t *__restrict__ seg_off, uint32_t *__restrict__ seg_cnt,
                                                  int32_t *__restrict__ event_point, int *overflow,
                                                  unsigned char *order_scratch) {
    const uint32_t lo = win_lo[s], n = win_hi[s] - lo, base = win_base[s];
    if (n == 0) {
        if (lo_excl == 0 && threadIdx.x == 0) {
            seg_off[2 * s] = base < cap_points ? base : 0;
            seg_off[2 * s + 1] = seg_off[2 * s];
            seg_cnt[2 * s] = 0;
            seg_cnt[2 * s + 1] = 0;
        }
        return;
    }
    if (n <= lo_excl || n > (uint32_t) CAP) return;
    if ((uint64_t) base + n > cap_points) {  // caller's buffers too small: report, emit empty segments
        if (threadIdx.x == 0) {
            *overflow = 1;
            seg_off[2 * s] = seg_off[2 * s + 1] = 0;
            seg_cnt[2 * s] = seg_cnt[2 * s + 1] = 0;
        }
        return;
    }
    using L = SliceLayout<CAP>;
    SliceWork<uint16_t> w;
    w.pts = reinterpret_cast<double2 *>(smem + L::pts_off);
    w.bend = reinterpret_cast<uint32_t *>(smem + L::bend_off);
    w.bend_words = CAP + 4;
    w.pts_words = 4 * CAP;
    w.pos = reinterpret_cast<uint32_t *>(smem + L::pos_off);
    w.sorted = reinterpret_cast<uint16_t *>(smem + L::sorted_off);
    w.rep = reinterpret_cast<uint16_t *>(smem + L::rep_off);
    w.pol = reinterpret_cast<uint8_t *>(smem + L::pol_off);
    w.red = reinterpret_cast<uint32_t *>(smem + L::red_off);
    uint32_t nP, nN;
    OrderScratch ord;
    uint32_t *evk = nullptr;
    if (order_scratch) {   // this workgroup's slice of the order scratch (order_scratch_bytes(CAP) each)
        unsigned char *o = order_scratch + (size_t) blockIdx.x * order_scratch_bytes(CAP);
        order_scratch_carve(o, CAP, &ord, &evk);
    }
    slice_window<T, false, uint16_t>(w, rec, lo, n, Log2c<NB>::value, xy_out + 2 * (size_t) base, event_point ? event_point + base : nullptr,
                                     &nP, &nN, order_scratch ? &ord : nullptr, evk);
    if (threadIdx.x == 0) {
        seg_off[2 * s] = base;
        seg_cnt[2 * s] = nP;
        seg_off[2 * s + 1] = base + nP;
        seg_cnt[2 * s + 1] = nN;
    }
}

// todo == nullptr: workgroup b handles window b (gridDim.x == S).  Otherwise the workgroups share the list of
// windows the pixel kernel left over (todo[0 .. *todo_count)), usually empty.
template <int CAP, int NB, int T>
__global__ __launch_bounds__(T) void slice_lds_kernel(const uint8_t *__restrict__ rec,
                                                      const uint32_t *__restrict__ win_lo,
                                                      const uint32_t *__restrict__ win_hi,
                                                      const uint32_t *__restrict__ win_base, uint32_t lo_excl,
                                                      uint32_t cap_points, double *__restrict__ xy_out,
                                                      uint32_t *__restrict__ seg_off, uint32_t *__restrict__ seg_cnt,
                                                      int32_t *__restrict__ event_point, int *overflow, uint32_t S,
                                                      const uint32_t *__restrict__ todo,
                                                      const uint32_t *__restrict__ todo_count,
                                                      unsigned char *order_scratch /* null: canonical order */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t count = todo ? *todo_count : S;
    for (uint32_t k = blockIdx.x; k < count; k += gridDim.x) {
        slice_tier_window<CAP, NB, T>(smem, todo ? todo[k] : k, rec, win_lo, win_hi, win_base, lo_excl, cap_points, xy_out,
                                      seg_off, seg_cnt, event_point, overflow, order_scratch);
        __syncthreads();
    }
}

// Bucket numbers of every sensor pixel the first pass can see (x <= 2047, y <= 1023), for the seven bucket counts a set of
// up to 1109 keys goes through: [pix = x << 10 | y] -> { epochs 4, 5, 6 packed 9 + 10 + 11 bits, epochs 0 .. 3 packed
// 4 + 5 + 6 + 7 bits }.  Built once per context (16 MB); the slicer then needs ONE 8-byte gather per key instead of the
// 64-bit hash (two table loads + hash_combine) and seven exact divisions.
__global__ void bucket_table_kernel(uint2 *__restrict__ tab) {
    const uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= (1u << 21)) return;
    const uint64_t h = ref_hash_combine2(HASH_INT.v[pix >> 10], HASH_INT.v[pix & 0x3FFu]);   // utility.hpp:38-51
    uint2 w;
    w.x = (uint32_t) (h % 257ull) | ((uint32_t) (h % 541ull) << 9) | ((uint32_t) (h % 1109ull) << 19);
    w.y = (uint32_t) (h % 13ull) | ((uint32_t) (h % 29ull) << 4) | ((uint32_t) (h % 59ull) << 9) | ((uint32_t) (h % 127ull) << 15);
    tab[pix] = w;
}


// first pass: workgroup b handles window b; what it cannot take goes to todo / todo_count
#ifndef ECAL_RO_WAVES
#define ECAL_RO_WAVES 5   // waves per SIMD the reference-order kernel is compiled for (0: the compiler's choice = 4; measured 50 M events: 4 -> 1.15 ms, 5 -> 1.03 ms, 6 (104 B of scratch per lane) -> 1.13 ms)
#endif
#if ECAL_RO_WAVES
#define ECAL_RO_ATTR __attribute__((amdgpu_waves_per_eu(ECAL_RO_WAVES, ECAL_RO_WAVES)))
#else
#define ECAL_RO_ATTR
#endif
__global__ __launch_bounds__(PXH_T) ECAL_RO_ATTR void slice_hash_ref_kernel(const uint8_t *__restrict__ rec,
                                                           const uint32_t *__restrict__ win_lo,
                                                           const uint32_t *__restrict__ win_hi,
                                                           const uint32_t *__restrict__ win_base, uint32_t cap_points,
                                                           double *__restrict__ xy_out, uint32_t *__restrict__ seg_off,
                                                           uint32_t *__restrict__ seg_cnt,
                                                           int32_t *__restrict__ event_point, int *overflow,
                                                           uint32_t *__restrict__ todo,
                                                           uint32_t *__restrict__ todo_count,
                                                           const uint2 *__restrict__ bucket_tab, uint32_t *__restrict__ xy16,
                                                           uint32_t *__restrict__ seg_fmt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    slice_hash_window<11, true>(smem, blockIdx.x, rec, win_lo, win_hi, win_base, cap_points, xy_out, seg_off, seg_cnt, event_point, overflow,
                                todo, todo_count, bucket_tab, xy16, seg_fmt);
}

__global__ __launch_bounds__(PXH_T) void slice_hash_kernel(const uint8_t *__restrict__ rec,
                                                           const uint32_t *__restrict__ win_lo,
                                                           const uint32_t *__restrict__ win_hi,
                                                           const uint32_t *__restrict__ win_base, uint32_t cap_points,
                                                           double *__restrict__ xy_out, uint32_t *__restrict__ seg_off,
                                                           uint32_t *__restrict__ seg_cnt,
                                                           int32_t *__restrict__ event_point, int *overflow,
                                                           uint32_t *__restrict__ todo,
                                                           uint32_t *__restrict__ todo_count, uint32_t *__restrict__ xy16,
                                                           uint32_t *__restrict__ seg_fmt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    slice_hash_window<11, false>(smem, blockIdx.x, rec, win_lo, win_hi, win_base, cap_points, xy_out, seg_off, seg_cnt, event_point, overflow,
                                 todo, todo_count, nullptr, xy16, seg_fmt);
}

// The LATENCY form of the two hash passes in the reference order (ecal_ctx::latency_pass: few windows hold work, a pass's time is
// the sum of its launches' single-workgroup latencies): workgroup b takes window b through the pass its event count asks for —
// a window of 2048 .. 4095 events does not wait for the first pass's launch to drain.  Same device code per window, same results;
// a window the first-pass code cannot take for another reason (more than 1109 keys of a polarity, a non-pixel coordinate) still
// goes to ITS list, which the second pass behind this launch works off (and the second pass's leftovers the third).
// (THIRD = false is the form of the search's EARLY passes — thousands of windows, throughput again —: compiled for three workgroups
// per CU, which its 52 KB of LDS allow; the third pass's windows then go through slice_hash_third_kernel behind it)
template <bool THIRD /* windows of 4096 .. 5119 events go through the third pass's code in this launch too */>
__global__ __launch_bounds__(PXH_T) __attribute__((amdgpu_waves_per_eu(THIRD ? 2 : 3, THIRD ? 2 : 3))) void slice_hash_ref_both_kernel(const uint8_t *__restrict__ rec, const uint32_t *__restrict__ win_lo,
                                                                    const uint32_t *__restrict__ win_hi, const uint32_t *__restrict__ win_base,
                                                                    uint32_t cap_points, double *__restrict__ xy_out, uint32_t *__restrict__ seg_off,
                                                                    uint32_t *__restrict__ seg_cnt, int32_t *__restrict__ event_point, int *overflow,
                                                                    uint32_t *__restrict__ todo, uint32_t *__restrict__ todo_count,
                                                                    uint32_t *__restrict__ todo2, uint32_t *__restrict__ todo2_count,
                                                                    uint32_t *__restrict__ todo3, uint32_t *__restrict__ todo3_count,
                                                                    const uint2 *__restrict__ bucket_tab, uint32_t *__restrict__ xy16,
                                                                    uint32_t *__restrict__ seg_fmt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t s = blockIdx.x;
    const uint32_t n = win_hi[s] - win_lo[s];
    if constexpr (THIRD) {
        // ONE launch for all three passes: a window the pass of its size cannot take (a set of more keys than that pass holds) goes
        // straight on to the next pass in this workgroup; only what the third cannot take either is listed (for the general tiers) —
        // the lists of the first two stay empty and their launches are not made
        bool done = false;
        if (n <= PixHash<11>::CAP)
            done = slice_hash_window<11, true>(smem, s, rec, win_lo, win_hi, win_base, cap_points, xy_out, seg_off, seg_cnt, event_point, overflow,
                                               nullptr, nullptr, bucket_tab, xy16, seg_fmt);
        if (!done && n <= PixHash<12>::CAP) {
            __syncthreads();
            done = slice_hash_window<12, true>(smem, s, rec, win_lo, win_hi, win_base, cap_points, xy_out, seg_off, seg_cnt, event_point, overflow,
                                               nullptr, nullptr, bucket_tab, xy16, seg_fmt);
        }
        if (!done) {
            __syncthreads();
            slice_hash_window<13, true>(smem, s, rec, win_lo, win_hi, win_base, cap_points, xy_out, seg_off, seg_cnt, event_point, overflow, todo3,
                                        todo3_count, bucket_tab, xy16, seg_fmt);
        }
        return;
    }
    if (n <= PixHash<11>::CAP)
        slice_hash_window<11, true>(smem, s, rec, win_lo, win_hi, win_base, cap_points, xy_out, seg_off, seg_cnt, event_point, overflow, todo,
                                    todo_count, bucket_tab, xy16, seg_fmt);
    else
        slice_hash_window<12, true>(smem, s, rec, win_lo, win_hi, win_base, cap_points, xy_out, seg_off, seg_cnt, event_point, overflow, todo2,
                                    todo2_count, bucket_tab, xy16, seg_fmt);
}

// third pass (reference order): the workgroups share the list the second pass left (windows of more than 4095 events, sets of more
// than 2048 keys); what it cannot take either (more than 5119 events, a set beyond the eighth epoch's 2357 buckets, a non-pixel
// coordinate) goes on to the general tiers' list
__global__ __launch_bounds__(PXH_T) __attribute__((amdgpu_waves_per_eu(2, 2))) void slice_hash_third_kernel(const uint8_t *__restrict__ rec, const uint32_t *__restrict__ win_lo,
                                                                 const uint32_t *__restrict__ win_hi, const uint32_t *__restrict__ win_base,
                                                                 uint32_t cap_points, double *__restrict__ xy_out, uint32_t *__restrict__ seg_off,
                                                                 uint32_t *__restrict__ seg_cnt, int32_t *__restrict__ event_point, int *overflow,
                                                                 uint32_t *__restrict__ todo, uint32_t *__restrict__ todo_count,
                                                                 const uint32_t *__restrict__ in_list, const uint32_t *__restrict__ in_count,
                                                                 const uint2 *__restrict__ bucket_tab, uint32_t *__restrict__ xy16,
                                                                 uint32_t *__restrict__ seg_fmt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t count = *in_count;
    for (uint32_t k = blockIdx.x; k < count; k += gridDim.x) {
        slice_hash_window<13, true>(smem, in_list[k], rec, win_lo, win_hi, win_base, cap_points, xy_out, seg_off, seg_cnt, event_point,
                                    overflow, todo, todo_count, bucket_tab, xy16, seg_fmt);
        __syncthreads();
    }
}

// second pass: the workgroups share the list the first pass left (in_list[0 .. *in_count)); windows of up to 4095 events
template <bool REFORDER>
#ifndef ECAL_SL2_WAVES
#define ECAL_SL2_WAVES 3
#endif
__global__ __launch_bounds__(PXH_T) __attribute__((amdgpu_waves_per_eu(ECAL_SL2_WAVES, ECAL_SL2_WAVES))) void slice_hash_list_kernel(const uint8_t *__restrict__ rec,
                                                                const uint32_t *__restrict__ win_lo,
                                                                const uint32_t *__restrict__ win_hi,
                                                                const uint32_t *__restrict__ win_base, uint32_t cap_points,
                                                                double *__restrict__ xy_out, uint32_t *__restrict__ seg_off,
                                                                uint32_t *__restrict__ seg_cnt,
                                                                int32_t *__restrict__ event_point, int *overflow,
                                                                uint32_t *__restrict__ todo, uint32_t *__restrict__ todo_count,
                                                                const uint32_t *__restrict__ in_list,
                                                                const uint32_t *__restrict__ in_count,
                                                                const uint2 *__restrict__ bucket_tab, uint32_t *__restrict__ xy16,
                                                                uint32_t *__restrict__ seg_fmt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t count = *in_count;
    for (uint32_t k = blockIdx.x; k < count; k += gridDim.x) {
        slice_hash_window<12, REFORDER>(smem, in_list[k], rec, win_lo, win_hi, win_base, cap_points, xy_out, seg_off, seg_cnt, event_point,
                              overflow, todo, todo_count, bucket_tab, xy16, seg_fmt);
        __syncthreads();
    }
}

constexpr int SLICE_BIG_T = 1024;

__device__ __forceinline__ void slice_big_window(const uint32_t s, unsigned long long *red64,
    const uint8_t *__restrict__ rec, const uint32_t *__restrict__ win_lo, const uint32_t *__restrict__ win_hi,
    const uint32_t *__restrict__ win_base, uint32_t lo_excl, uint32_t cap_points, double *__restrict__ xy_out,
    uint32_t *__restrict__ seg_off, uint32_t *__restrict__ seg_cnt, int32_t *__restrict__ event_point, int *overflow,
    double2 *g_pts, uint8_t *g_pol, uint32_t *g_bend, uint32_t *g_sorted, uint32_t *g_rep, uint32_t *g_pos,
    unsigned char *order_scratch /* null: canonical order; else order_scratch_bytes(cap_points) + 64 S bytes */) {
    const uint32_t lo = win_lo[s], n = win_hi[s] - lo, base = win_base[s];
    if (n <= lo_excl) return;
    if ((uint64_t) base + n > cap_points || n >= 0x80000000u) {
        if (threadIdx.x == 0) {
            *overflow = 1;
            seg_off[2 * s] = seg_off[2 * s + 1] = 0;
            seg_cnt[2 * s] = seg_cnt[2 * s + 1] = 0;
        }
        return;
    }
    SliceWork<uint32_t> w;
    w.pts = g_pts + base;
    w.pol = g_pol + base;
    w.bend = g_bend + base + s;  // nb + 1 <= n + 1 words per window
    w.sorted = g_sorted + base;
    w.rep = g_rep + base;
    w.pos = g_pos + base;
    w.red = reinterpret_cast<uint32_t *>(red64);
    uint32_t nb_log = 31u - (uint32_t) __clz((int) n);  // largest power of two <= n
    if (nb_log > 20u) nb_log = 20u;
    uint32_t nP, nN;
    OrderScratch ord;
    uint32_t *evk = nullptr;
    if (order_scratch) {   // per-window slices: the n-entry arrays at `base`, fa at 9 base / 4 + 16 s (it holds B(n) < 9 n / 4 + 16 words)
        const size_t cap = cap_points;
        order_scratch_carve(order_scratch, cap, &ord, &evk);
        ord.h += base;
        ord.cur += base;
        ord.slot += base;
        ord.cnt += base;
        ord.bas += base;
        ord.region += base;
        evk += base;
        ord.fa += (size_t) base * 9 / 4 + 16 * (size_t) s;
    }
    slice_window<SLICE_BIG_T, true, uint32_t>(w, rec, lo, n, nb_log, xy_out + 2 * (size_t) base, event_point ? event_point + base : nullptr, &nP,
                                              &nN, order_scratch ? &ord : nullptr, evk);
    if (threadIdx.x == 0) {
        seg_off[2 * s] = base;
        seg_cnt[2 * s] = nP;
        seg_off[2 * s + 1] = base + nP;
        seg_cnt[2 * s + 1] = nN;
    }
}


// the windows beyond the LDS tiers, from the to-do list the earlier passes left (all S windows without one): a small grid walks
// the list — a launch of S workgroups of 1024 threads that find nothing to do cost 30 - 60 us per call when the sizes are unknown
__global__ __launch_bounds__(SLICE_BIG_T) void slice_big_kernel(
    const uint8_t *__restrict__ rec, const uint32_t *__restrict__ win_lo, const uint32_t *__restrict__ win_hi,
    const uint32_t *__restrict__ win_base, uint32_t lo_excl, uint32_t cap_points, double *__restrict__ xy_out,
    uint32_t *__restrict__ seg_off, uint32_t *__restrict__ seg_cnt, int32_t *__restrict__ event_point, int *overflow,
    double2 *g_pts, uint8_t *g_pol, uint32_t *g_bend, uint32_t *g_sorted, uint32_t *g_rep, uint32_t *g_pos,
    unsigned char *order_scratch, uint32_t S, const uint32_t *__restrict__ todo, const uint32_t *__restrict__ todo_count,
    uint32_t *seen /* ecal_ctx::tail_seen: what the stage's lists held (cnt_a, cnt_b: their counters, null = none) */,
    const uint32_t *cnt_a, const uint32_t *cnt_b) {
    __shared__ unsigned long long red64[17];
    const uint32_t n_work = todo ? *todo_count : S;
    if (seen && blockIdx.x == 0 && threadIdx.x == 0) {
        seen[0] = cnt_a ? *cnt_a : 0u;
        seen[1] = cnt_b ? *cnt_b : 0u;
    }
    for (uint32_t k = blockIdx.x; k < n_work; k += gridDim.x) {
        slice_big_window(todo ? todo[k] : k, red64, rec, win_lo, win_hi, win_base, lo_excl, cap_points, xy_out, seg_off, seg_cnt, event_point,
                         overflow, g_pts, g_pol, g_bend, g_sorted, g_rep, g_pos, order_scratch);
        __syncthreads();
    }
}

}  // namespace ecal

using namespace ecal;

static constexpr int SCAP0 = 2048, SCAP1 = 5120;

extern "C" int ecal_window_bounds_dev(ecal_ctx *ctx, const uint8_t *d_events, uint64_t n_events, const double *d_t0,
                                      const double *d_t1, uint32_t S, uint32_t *d_win_lo, uint32_t *d_win_hi,
                                      uint32_t *d_win_base, void *stream) {
    const ecal_range range__(ctx, "ecal_window_bounds");
    if (!ctx) return ECAL_ERR_INVALID;
    if (n_events > 0xFFFFFFFFull) {
        ctx->last_error = "more than 2^32-1 events in one stream";
        return ECAL_ERR_RANGE;
    }
    if (S == 0) return ECAL_OK;
    if ((n_events && !d_events) || !d_t0 || !d_t1 || !d_win_lo || !d_win_hi || !d_win_base) {
        ctx->last_error = "null pointer";
        return ECAL_ERR_INVALID;
    }
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t) stream;
    const uint32_t n_wg = (S + WB_T - 1) / WB_T;
    // the look-back table and its epoch belong to ONE stream of the context (the first that asks): calls in flight on two
    // streams would overwrite each other's status words with another epoch and a workgroup looking back would wait for ever
    // — a call on any other stream takes the two-kernel form, which uses no context scratch
    // The binding moves when the caller's stream changes and the owner has nothing in flight (an event behind every fused call):
    // a context whose first call came from a warm-up stream is not stuck on the two-kernel form for life.
    if (!ctx->wb_stream_set) {
        ctx->wb_stream = st;
        ctx->wb_stream_set = true;
    } else if (ctx->wb_stream != st && ctx->wb_done && hipEventQuery(ctx->wb_done) == hipSuccess) {
        ctx->wb_stream = st;
    }
    uint32_t *ticket = (ctx->sw.bounds_two_kernels || ctx->wb_stream != st) ? nullptr : ecal_zero_words(ctx, st, 1);
    if (ticket && ctx->wb_status.cap < (size_t) n_wg * sizeof(unsigned long long)) {
        int rc;
        if ((rc = ecal_ensure(ctx, ctx->wb_status, (size_t) n_wg * sizeof(unsigned long long)))) return rc;
        // fresh memory must not carry a word that reads as this call's: wipe it once, the epochs do the rest
        ECAL_HIP_TRY(ctx, hipMemsetAsync(ctx->wb_status.ptr, 0, ctx->wb_status.cap, st));
    }
    if (ticket) {
        ctx->wb_epoch = (ctx->wb_epoch % 0x3FFFFFFFu) + 1u;       // 1 .. 2^30 - 1, never the wiped table's 0
        hipLaunchKernelGGL(window_bounds_base_kernel, dim3(n_wg), dim3(WB_T), 0, st, d_events, n_events, d_t0, d_t1, S, d_win_lo, d_win_hi,
                           d_win_base, (unsigned long long *) ctx->wb_status.ptr, ctx->wb_epoch, ticket);
        if (!ctx->wb_done && hipEventCreateWithFlags(&ctx->wb_done, hipEventDisableTiming) != hipSuccess) ctx->wb_done = nullptr;
        if (ctx->wb_done) (void) hipEventRecord(ctx->wb_done, st);
    } else {   // (no zeroed word to be had, or the debug switch: the search and the scan as two launches)
        hipLaunchKernelGGL(window_bounds_kernel, dim3((S + 255) / 256), dim3(256), 0, st, d_events, n_events, d_t0, d_t1, S,
                           d_win_lo, d_win_hi);
        hipLaunchKernelGGL(window_base_kernel, dim3(1), dim3(1024), 0, st, d_win_lo, d_win_hi, S, d_win_base);
    }
    ECAL_HIP_TRY(ctx, hipGetLastError());
    return ECAL_OK;
}

extern "C" int ecal_check_sorted_dev(ecal_ctx *ctx, const uint8_t *d_events, uint64_t n_events, int *d_flag,
                                     void *stream) {
    if (!ctx || !d_flag || (n_events && !d_events)) return ECAL_ERR_INVALID;
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t) stream;
    ECAL_HIP_TRY(ctx, hipMemsetAsync(d_flag, 0, sizeof(int), st));
    if (n_events > 1) {
        const uint64_t blocks = (n_events - 1 + 255) / 256;
        hipLaunchKernelGGL(check_sorted_kernel, dim3((uint32_t) blocks), dim3(256), 0, st, d_events, n_events, d_flag);
        ECAL_HIP_TRY(ctx, hipGetLastError());
    }
    return ECAL_OK;
}

int ecal_ensure_bucket_table(ecal_ctx *ctx, hipStream_t st) {
    if (ctx->bucket_tab_built) return ECAL_OK;   // once per context (16 MB, ~0.1 ms)
    int rc;
    if ((rc = ecal_ensure(ctx, ctx->bucket_tab, sizeof(uint2) << 21))) return rc;
    hipLaunchKernelGGL(bucket_table_kernel, dim3((1u << 21) / 256), dim3(256), 0, st, (uint2 *) ctx->bucket_tab.ptr);
    ECAL_HIP_TRY(ctx, hipStreamSynchronize(st));   // (later calls may come on other streams)
    ctx->bucket_tab_built = true;
    return ECAL_OK;
}

// ---- packed points (ecal_packed_points): segments of integer pixels as x | y << 16, doubles on request -------------------------
namespace {
// the segments list[0 .. *count) (or all S when list == nullptr) that exist packed only get their doubles written (fmt 1 -> 3)
__global__ __launch_bounds__(256) void unpack_segments_kernel(const uint32_t *__restrict__ list, const uint32_t *__restrict__ count, uint32_t S,
                                                             const uint32_t *__restrict__ seg_off, const uint32_t *__restrict__ seg_cnt,
                                                             const uint32_t *__restrict__ xy16, uint32_t *__restrict__ fmt,
                                                             double *__restrict__ xy, int windows) {
    const uint32_t n_work = list ? *count : S;
    for (uint32_t k = blockIdx.x; k < n_work; k += gridDim.x) {
        const uint32_t e = list ? list[k] : k;
        for (int h = 0; h < (windows ? 2 : 1); h++) {   // (a list of windows names both of their segments)
            const uint32_t s = windows ? 2u * (e & 0x3FFFFFFFu) + (uint32_t) h : e;
            if (fmt[s] != 1u) continue;
            const uint32_t o = seg_off[s], n = seg_cnt[s];
            double2 *out = reinterpret_cast<double2 *>(xy) + o;
            for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
                const uint32_t w = xy16[o + i];
                out[i] = make_double2((double) (int) (short) (w & 0xFFFFu), (double) (((int) w) >> 16));
            }
            __syncthreads();
            if (threadIdx.x == 0) fmt[s] = 3u;
            __syncthreads();
        }
    }
}
}  // namespace

// the doubles of the listed segments (list == nullptr: of all S segments; windows != 0: the list names windows, S counts segments)
int ecal_unpack_listed(ecal_ctx *ctx, const ecal_packed_points *pk, const uint32_t *d_list, const uint32_t *d_count, uint32_t S,
                       const uint32_t *d_seg_off, const uint32_t *d_seg_cnt, double *d_xy, int windows, hipStream_t st) {
    if (!pk || !pk->d_xy16 || !pk->d_seg_fmt || S == 0) return ECAL_OK;
    const uint32_t grid = S < 1024u ? S : 1024u;
    hipLaunchKernelGGL(unpack_segments_kernel, dim3(grid), dim3(256), 0, st, d_list, d_count, S, d_seg_off, d_seg_cnt,
                       (const uint32_t *) pk->d_xy16, pk->d_seg_fmt, d_xy, windows);
    ECAL_HIP_TRY(ctx, hipGetLastError());
    return ECAL_OK;
}

extern "C" int ecal_unpack_points_dev(ecal_ctx *ctx, const ecal_packed_points *pk, const uint32_t *d_seg_off, const uint32_t *d_seg_cnt,
                                      uint32_t n_segments, double *d_xy, void *stream) {
    if (!ctx || !pk || !d_seg_off || !d_seg_cnt || !d_xy) return ECAL_ERR_INVALID;
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return ecal_unpack_listed(ctx, pk, nullptr, nullptr, n_segments, d_seg_off, d_seg_cnt, d_xy, 0, (hipStream_t) stream);
}

extern "C" int ecal_slice_events_dev(ecal_ctx *ctx, const uint8_t *d_events, uint64_t n_events,
                                     const uint32_t *d_win_lo, const uint32_t *d_win_hi, const uint32_t *d_win_base,
                                     uint32_t S, uint32_t max_win_events, uint32_t cap_points, double *d_xy,
                                     uint32_t *d_seg_off, uint32_t *d_seg_cnt, int32_t *d_event_point, int *d_overflow,
                                     void *stream) {
    return ecal_slice_events_packed_dev(ctx, d_events, n_events, d_win_lo, d_win_hi, d_win_base, S, max_win_events, cap_points, d_xy,
                                        d_seg_off, d_seg_cnt, d_event_point, d_overflow, nullptr, stream);
}

extern "C" int ecal_slice_events_packed_dev(ecal_ctx *ctx, const uint8_t *d_events, uint64_t n_events,
                                            const uint32_t *d_win_lo, const uint32_t *d_win_hi, const uint32_t *d_win_base,
                                            uint32_t S, uint32_t max_win_events, uint32_t cap_points, double *d_xy,
                                            uint32_t *d_seg_off, uint32_t *d_seg_cnt, int32_t *d_event_point, int *d_overflow,
                                            const ecal_packed_points *pk, void *stream) {
    const ecal_range range__(ctx, "ecal_slice_events");
    if (!ctx) return ECAL_ERR_INVALID;
    // new points go into the segment arrays: the kd-trees the pixel DBSCAN kernel exported for the member-order kernel
    // (ecal_ctx::px_tree) belong to the points of ITS call — the next DBSCAN call exports its own
    ctx->px_tree_labels = nullptr;
    if (S == 0) return ECAL_OK;
    if (pk && (!pk->d_xy16 || !pk->d_seg_fmt)) pk = nullptr;
    uint32_t *const xy16 = pk ? pk->d_xy16 : nullptr, *const sfmt = pk ? pk->d_seg_fmt : nullptr;
    if ((n_events && !d_events) || !d_win_lo || !d_win_hi || !d_win_base || !d_seg_off || !d_seg_cnt || !d_overflow ||
        (cap_points && !d_xy)) {   // (d_event_point may be null: the event -> point map is not wanted)
        ctx->last_error = "null pointer";
        return ECAL_ERR_INVALID;
    }
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t) stream;
    const bool reforder = ctx->point_order == ECAL_ORDER_REFERENCE;
    constexpr size_t H11 = PixHash<11>::bytes > PixHash<11>::obytes ? PixHash<11>::bytes : PixHash<11>::obytes;
    constexpr size_t H12 = PixHash<12>::bytes > PixHash<12>::obytes ? PixHash<12>::bytes : PixHash<12>::obytes;
    constexpr size_t H13 = PixHash<13>::bytes > PixHash<13>::obytes ? PixHash<13>::bytes : PixHash<13>::obytes;
    if (!ctx->slice_attrs_set) {
        ECAL_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&slice_lds_kernel<SCAP0, 2048, 256>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize,
                                              (int) SliceLayout<SCAP0>::bytes));
        ECAL_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&slice_lds_kernel<SCAP1, 4096, 512>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize,
                                              (int) SliceLayout<SCAP1>::bytes));
        ECAL_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&slice_hash_list_kernel<false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int) PixHash<12>::bytes));
        ECAL_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&slice_hash_list_kernel<true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int) H12));
        ECAL_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&slice_hash_ref_both_kernel<false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int) H12));
        ECAL_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&slice_hash_ref_both_kernel<true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int) H13));
        ECAL_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&slice_hash_third_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int) H13));
        ctx->slice_attrs_set = true;
    }
    if (d_overflow != ctx->overflow_sticky) ECAL_HIP_TRY(ctx, hipMemsetAsync(d_overflow, 0, sizeof(int), st));   // (ecal_ctx::overflow_sticky)
    // (packed points: every segment starts as "doubles"; the pixel kernels mark the windows they pack.  The hash slicers — the
    // default — write the mark of EVERY window they look at, also of the ones they pass on: no wipe, one launch less per pass)
    const bool hash_slicer = !ctx->sw.slice_no_pixel;
    if (sfmt && !hash_slicer) ECAL_HIP_TRY(ctx, hipMemsetAsync(sfmt, 0, 2 * (size_t) S * sizeof(uint32_t), st));
    const uint32_t mx = max_win_events ? max_win_events : 0xFFFFFFFFu;
    // pixel windows first; what they leave over (longer windows, non-integer coordinates) is listed for the general tiers
    const uint32_t *todo = nullptr, *todo_count = nullptr;
    uint32_t grid = S;
    // lean: the listed windows (none, when this stage last ran) all go to the global-scratch tier, one launch behind the first
    // pass instead of four (ecal_ctx::tail_seen); only without size hints — a caller who names its sizes gets what it asks for
    const int plan = hash_slicer && max_win_events == 0 ? ecal_tail_plan(ctx, ECAL_TAIL_SLICE) : ECAL_PLAN_TIERED;
    const bool lean = plan == ECAL_PLAN_LEAN;
    // semi: first and second pass as always, then the global-scratch tier alone for whatever the second pass leaves (nothing, when
    // this stage last ran) instead of the two LDS tiers + it
    const bool semi = plan == ECAL_PLAN_SEMI && reforder;
    const uint32_t *cnt_a = nullptr, *cnt_b = nullptr;
    if (!ctx->sw.slice_no_pixel) {
        int rc;
        if ((rc = ecal_ensure(ctx, ctx->pxs_todo, (3 * (size_t) S + 8) * sizeof(uint32_t)))) return rc;
        uint32_t *cnt = (uint32_t *) ctx->pxs_todo.ptr, *list = cnt + 8, *cnt2 = cnt + 1, *list2 = list + S, *cnt3 = cnt + 2, *list3 = list2 + S;
        // the lists' counters: words that are zero already, else three wiped now
        if (uint32_t *z = ecal_zero_words(ctx, st, 3)) {
            cnt = z;
            cnt2 = z + 1;
            cnt3 = z + 2;
        } else {
            ECAL_HIP_TRY(ctx, hipMemsetAsync(cnt, 0, 3 * sizeof(uint32_t), st));
        }
        todo = list;
        todo_count = cnt;
        cnt_a = cnt;
        if (reforder) {
            if ((rc = ecal_ensure_bucket_table(ctx, st))) return rc;
            // (few windows at work: a window goes through the pass its size asks for in ONE launch; very few — latency_pass 2, the
            // tail of the keyframe search —: the third pass's windows as well, in workgroups of 66 KB instead of 52)
            if (ecal_latency_level(ctx) >= 2 && !lean)
                hipLaunchKernelGGL(slice_hash_ref_both_kernel<true>, dim3(S), dim3(PXH_T), H13, st, d_events, d_win_lo, d_win_hi, d_win_base,
                                   cap_points, d_xy, d_seg_off, d_seg_cnt, d_event_point, d_overflow, list, cnt, list2, cnt2, list3, cnt3,
                                   (const uint2 *) ctx->bucket_tab.ptr, xy16, sfmt);
            else if (ecal_latency_level(ctx) && !lean)
                hipLaunchKernelGGL(slice_hash_ref_both_kernel<false>, dim3(S), dim3(PXH_T), H12, st, d_events, d_win_lo, d_win_hi, d_win_base,
                                   cap_points, d_xy, d_seg_off, d_seg_cnt, d_event_point, d_overflow, list, cnt, list2, cnt2, list3, cnt3,
                                   (const uint2 *) ctx->bucket_tab.ptr, xy16, sfmt);
            else
            hipLaunchKernelGGL(slice_hash_ref_kernel, dim3(S), dim3(PXH_T), H11, st, d_events, d_win_lo, d_win_hi,
                               d_win_base, cap_points, d_xy, d_seg_off, d_seg_cnt, d_event_point, d_overflow, list, cnt,
                               (const uint2 *) ctx->bucket_tab.ptr, xy16, sfmt);
        }
        else
            hipLaunchKernelGGL(slice_hash_kernel, dim3(S), dim3(PXH_T), PixHash<11>::bytes, st, d_events, d_win_lo, d_win_hi,
                               d_win_base, cap_points, d_xy, d_seg_off, d_seg_cnt, d_event_point, d_overflow, list, cnt, xy16, sfmt);
        // (reference order: the second pass also takes the windows whose sets outgrow the first pass's bucket tables)
        if (reforder && !lean && ecal_latency_level(ctx) >= 2) {   // (all three passes were that one launch: its only list is the general tiers')
            cnt_b = cnt3;
            todo = list3;
            todo_count = cnt3;
        } else if (!lean && (reforder || mx > PixHash<11>::CAP)) {
            cnt_b = cnt2;
            const uint32_t grid2 = S < 768u ? S : 768u;
            if (reforder)
                hipLaunchKernelGGL(slice_hash_list_kernel<true>, dim3(grid2), dim3(PXH_T), H12, st, d_events, d_win_lo,
                                   d_win_hi, d_win_base, cap_points, d_xy, d_seg_off, d_seg_cnt, d_event_point, d_overflow, list2,
                                   cnt2, (const uint32_t *) list, (const uint32_t *) cnt, (const uint2 *) ctx->bucket_tab.ptr, xy16, sfmt);
            else
                hipLaunchKernelGGL(slice_hash_list_kernel<false>, dim3(grid2), dim3(PXH_T), PixHash<12>::bytes, st, d_events, d_win_lo,
                                   d_win_hi, d_win_base, cap_points, d_xy, d_seg_off, d_seg_cnt, d_event_point, d_overflow, list2,
                                   cnt2, (const uint32_t *) list, (const uint32_t *) cnt, (const uint2 *) nullptr, xy16, sfmt);
            todo = list2;
            todo_count = cnt2;
            if (reforder) {   // the third pass: what the second one leaves that is still a pixel window
                hipLaunchKernelGGL(slice_hash_third_kernel, dim3(S < 512u ? S : 512u), dim3(PXH_T), H13, st, d_events, d_win_lo, d_win_hi,
                                   d_win_base, cap_points, d_xy, d_seg_off, d_seg_cnt, d_event_point, d_overflow, list3, cnt3,
                                   (const uint32_t *) list2, (const uint32_t *) cnt2, (const uint2 *) ctx->bucket_tab.ptr, xy16, sfmt);
                cnt_b = cnt3;
                todo = list3;
                todo_count = cnt3;
            }
        }
        grid = S < 512u ? S : 512u;
    }
    // general tiers; in reference order they take their order scratch from global memory, one slice per workgroup
    unsigned char *ord_lds = nullptr, *ord_big = nullptr;
    if (reforder) {
        int rc;
        const size_t cap = mx > (uint32_t) SCAP0 ? SCAP1 : SCAP0;
        if ((rc = ecal_ensure(ctx, ctx->sl_order, (size_t) grid * order_scratch_bytes(cap)))) return rc;
        ord_lds = (unsigned char *) ctx->sl_order.ptr;
    }
    if (!lean && !semi)
    hipLaunchKernelGGL((slice_lds_kernel<SCAP0, 2048, 256>), dim3(grid), dim3(256), SliceLayout<SCAP0>::bytes, st, d_events,
                       d_win_lo, d_win_hi, d_win_base, 0u, cap_points, d_xy, d_seg_off, d_seg_cnt, d_event_point,
                       d_overflow, S, todo, todo_count, ord_lds);
    if (!lean && !semi && mx > (uint32_t) SCAP0)
        hipLaunchKernelGGL((slice_lds_kernel<SCAP1, 4096, 512>), dim3(grid), dim3(512), SliceLayout<SCAP1>::bytes, st,
                           d_events, d_win_lo, d_win_hi, d_win_base, (uint32_t) SCAP0, cap_points, d_xy, d_seg_off,
                           d_seg_cnt, d_event_point, d_overflow, S, todo, todo_count, ord_lds);
    if (mx > (uint32_t) SCAP1) {
        const size_t w = cap_points;
        int rc;
        if ((rc = ecal_ensure(ctx, ctx->sl_pts, w * sizeof(double2)))) return rc;
        if ((rc = ecal_ensure(ctx, ctx->sl_pol, w))) return rc;
        if ((rc = ecal_ensure(ctx, ctx->sl_bend, (w + S + 4) * sizeof(uint32_t)))) return rc;
        if ((rc = ecal_ensure(ctx, ctx->sl_sorted, w * sizeof(uint32_t)))) return rc;
        if ((rc = ecal_ensure(ctx, ctx->sl_rep, w * sizeof(uint32_t)))) return rc;
        if ((rc = ecal_ensure(ctx, ctx->sl_pos, w * sizeof(uint32_t)))) return rc;
        if (reforder) {
            if ((rc = ecal_ensure(ctx, ctx->sl_order_big, order_scratch_bytes(w) + 64 * ((size_t) S + 1)))) return rc;
            ord_big = (unsigned char *) ctx->sl_order_big.ptr;
        }
        hipLaunchKernelGGL(slice_big_kernel, dim3(S < 256u ? S : 256u), dim3(SLICE_BIG_T), 0, st, d_events, d_win_lo, d_win_hi, d_win_base,
                           lean || semi ? 0u : (uint32_t) SCAP1, cap_points, d_xy, d_seg_off, d_seg_cnt, d_event_point, d_overflow,
                           (double2 *) ctx->sl_pts.ptr, (uint8_t *) ctx->sl_pol.ptr, (uint32_t *) ctx->sl_bend.ptr,
                           (uint32_t *) ctx->sl_sorted.ptr, (uint32_t *) ctx->sl_rep.ptr, (uint32_t *) ctx->sl_pos.ptr, ord_big, S, todo,
                           todo_count, hash_slicer ? ctx->tail_seen_dev + ECAL_TAIL_SLICE : nullptr, cnt_a, cnt_b);
    }
    ECAL_HIP_TRY(ctx, hipGetLastError());
    return ECAL_OK;
}

extern "C" int ecal_set_point_order(ecal_ctx *ctx, int order) {
    if (!ctx || (order != ECAL_ORDER_REFERENCE && order != ECAL_ORDER_FIRST_OCCURRENCE)) return ECAL_ERR_INVALID;
    ctx->point_order = order;
    return ECAL_OK;
}

extern "C" int ecal_get_point_order(const ecal_ctx *ctx) { return ctx ? ctx->point_order : ECAL_ERR_INVALID; }

/* the tables of slice_order.hpp, for the CPU suite (tests/test_oracle_events.py checks them against libstdc++ itself) */
extern "C" uint64_t ecal_ref_bucket_step(int epoch) { return (epoch >= 0 && epoch < REF_N_STEPS) ? ref_bucket_step(epoch) : 0; }
extern "C" uint64_t ecal_ref_pixel_hash(double x, double y) {
    return ref_hash_combine2(ref_hash_f64_bits(__builtin_bit_cast(uint64_t, x)), ref_hash_f64_bits(__builtin_bit_cast(uint64_t, y)));
}

#ifdef ECAL_PHASE_PROF
extern "C" int ecal_debug_ro_cycles(unsigned long long *out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(ecal::g_ro_cycles), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(ecal::g_ro_cycles), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
