// The reference's element order of EventFrame's pixel sets, computed in parallel.
//
// positiveEvents_ / negativeEvents_ leave the EventFrame constructor in the iteration order of a
//   std::unordered_set<Eigen::Vector2d, EigenMatrixHash<Eigen::Vector2d>, std::equal_to<>, aligned_allocator>
// (event/src/EventFrame.cpp:12-13,34-35; hash = boost-style hash_combine of std::hash<double> over x then y,
// core/utility/include/opengv2/utility/utility.hpp:38-51).  The kd-tree of DBSCAN is built by inserting the points in
// that order (dbscan.h:186-196, kdtree.cpp:106-146) and its range query prunes strictly at exactly eps
// (kdtree.cpp:169), so cluster assignments depend on it: `.bin`-level parity needs the order itself.  It is an
// artefact of libstdc++'s _Hashtable (unique keys, hash code cached), restated here as rules on the singly linked
// node list; oracle/event_oracle.cpp runs the real container (oracle_event_frame_ref) and these rules
// (oracle_event_frame_model) side by side, tests/test_oracle_events.py requires them to agree:
//   * keys u = 0, 1, ... = the unique pixels of one polarity in order of first occurrence (a repeated insert is a no-op;
//     the +/- cancellation erases afterwards and keeps the relative order of what stays, EventFrame.cpp:24-32);
//   * bucket count: 13 while the set holds <= 13 keys; the insert that would exceed the bucket count B first rehashes
//     to the next prime of the library's list >= 2 B: 29, 59, 127, 257, 541, 1109, 2357, 5087, ...
//     (_Prime_rehash_policy::_M_need_rehash / _M_next_bkt, max_load_factor 1);
//   * insert with B buckets: if a node of bucket (hash % B) is in the list the new node goes in FRONT of that bucket's
//     run of nodes, else to the front of the whole list (_M_insert_bucket_begin);
//   * rehash: walk the list front to back and re-insert every node by the same rule (_M_rehash_aux, unique keys).
// Hence one "epoch" (constant B) maps a sequence — the old list front to back, then the epoch's new keys in arrival
// order — to the list   reverse( stable sort of the sequence by the first sequence position of the element's bucket ),
// i.e. position = n - 1 - (#elements in buckets first seen earlier + #same-bucket elements earlier in the sequence).
// Epoch sizes double, so the whole order costs about twice the last epoch.  Pinned to libstdc++ as shipped with
// g++ 11.4 (the image's; the same rules hold for every libstdc++ since the 4.9 hashtable, GCC 12+'s small-size path
// does not apply to a fast hash).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ecal {

// ---- std::hash<double> of libstdc++ (64-bit): 0 for +-0.0, else _Hash_bytes(&v, 8, 0xc70f6907) ------------------------
__host__ __device__ constexpr uint64_t ref_shift_mix(uint64_t v) { return v ^ (v >> 47); }
__host__ __device__ constexpr uint64_t ref_hash_f64_bits(uint64_t bits) {
    // bits of +0.0 / -0.0 hash to 0 (std::hash<double>::operator(): `__val != 0.0 ? hash(__val) : 0`)
    if ((bits << 1) == 0) return 0;
    const uint64_t mul = (0xc6a4a793ull << 32) + 0x5bd1e995ull;
    uint64_t h = 0xc70f6907ull ^ (8 * mul);
    const uint64_t data = ref_shift_mix(bits * mul) * mul;
    h ^= data;
    h *= mul;
    h = ref_shift_mix(h) * mul;
    h = ref_shift_mix(h);
    return h;
}
// EigenMatrixHash<Vector2d> (utility.hpp:38-51) from the two coordinate hashes
__host__ __device__ constexpr uint64_t ref_hash_combine2(uint64_t hx, uint64_t hy) {
    uint64_t seed = 0;
    seed ^= hx + 0x9e3779b9ull + (seed << 6) + (seed >> 2);
    seed ^= hy + 0x9e3779b9ull + (seed << 6) + (seed >> 2);
    return seed;
}
__device__ __forceinline__ uint64_t ref_pixel_hash(double x, double y) {
    return ref_hash_combine2(ref_hash_f64_bits((uint64_t) __double_as_longlong(x)),
                             ref_hash_f64_bits((uint64_t) __double_as_longlong(y)));
}

// ---- the bucket counts, epoch by epoch ------------------------------------------------------------------------------
// B_0 = _M_next_bkt(12), B_{e+1} = _M_next_bkt(2 B_e): entries of libstdc++'s __prime_list (src/shared/hashtable-aux.cc).
// tests/test_oracle_events.py checks the list against the library's own policy object.  Epoch e holds the keys
// u < B_e; 28 steps cover 2^32 - 1 events.
constexpr int REF_N_STEPS = 28;
__host__ __device__ constexpr uint64_t ref_bucket_step(int e) {
    constexpr uint64_t steps[REF_N_STEPS] = {13ull,        29ull,        59ull,        127ull,       257ull,       541ull,        1109ull,
                                             2357ull,      5087ull,      10273ull,     20753ull,     42043ull,     85229ull,      172933ull,
                                             351061ull,    712697ull,    1447153ull,   2938679ull,   5967347ull,   12117689ull,   24607243ull,
                                             49969847ull,  101473717ull, 206062531ull, 418451333ull, 849749479ull, 1725587117ull, 3504151727ull};
    return steps[e];
}
// number of epochs a set of m keys goes through (m >= 1): smallest E with m <= B_{E-1}
__host__ __device__ constexpr int ref_epochs(uint64_t m) {
    int e = 0;
    while (ref_bucket_step(e) < m) e++;
    return e + 1;
}

// ---- generic form: one polarity's list order in global (or any) memory ---------------------------------------------
// One workgroup.  In: h[u] = hash of key u (u = 0 .. m-1, first-occurrence order).  Out: cur[u] = position of key u in
// the set's iteration order (0 = begin()) before the cancellation.  Scratch (u32 unless noted): slot[m], cnt[m], bas[m],
// region[m], fa[B(m)] with B(m) < 9 m / 4 + 16.  Used by the general slicing tiers (any coordinates, any window size);
// the pixel kernels carry their own LDS-resident form (ecal_events.hip).
struct OrderScratch {
    uint64_t *h;
    uint32_t *cur, *slot, *cnt, *bas, *region, *fa;
};

template <int T>
__device__ __forceinline__ uint32_t block_exscan_u32(uint32_t v, uint32_t *red /* T/64 + 1 words */, uint32_t *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) red[wave] = inc;
    __syncthreads();
    uint32_t pre = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < T / 64; w++) {
        const uint32_t x = red[w];
        if (w < wave) pre += x;
        tot += x;
    }
    __syncthreads();
    *total = tot;
    return pre + inc - v;
}

// lds (optional): 8 lds_keys words of LDS (8-byte aligned) that are free during the call.  The epochs of up to lds_keys buckets
// — with 541, six of the nine a set of ~2500 keys goes through — then keep all their arrays (and the hashes of their keys)
// there: an epoch is six barrier-separated phases, each a round trip to its arrays, and from global memory those round trips
// are what the general slicing tiers' order pass consists of.
// h % B without the 64-bit division (a library call of ~200 instructions, asked four times per key and epoch before): for
// B < 2^16, h = hi 2^32 + lo gives (hi % B) (2^32 % B) + lo % B < B^2 <= 2^32 - 2^17 + 1 + 2^16: three 32-bit remainders.
__device__ __forceinline__ uint32_t ref_bucket_of(uint64_t h, uint64_t B, uint32_t two32_mod_b) {
    if (B < 65536ull) {
        const uint32_t b = (uint32_t) B;
        return ((uint32_t) (h >> 32) % b * two32_mod_b + (uint32_t) h % b) % b;
    }
    return (uint32_t) (h % B);
}

// KPT > 0: no thread holds more than KPT keys (m <= KPT T) — a key's bucket is then worked out once per epoch and kept in a
// register; KPT = 0: any m, the bucket is worked out where it is needed.
template <int T, int KPT = 0>
__device__ void reference_list_order(const OrderScratch w, uint32_t m, uint32_t *red, uint32_t *lds = nullptr, uint32_t lds_keys = 0) {
    const uint32_t tid = threadIdx.x;
    OrderScratch l = w;   // the small epochs' arrays: h (2 words a key), cur, slot, cnt, bas, region, fa
    if (lds_keys < 13u) lds = nullptr;
    if (lds) {
        l.h = reinterpret_cast<uint64_t *>(lds);
        l.cur = lds + 2 * lds_keys;
        l.slot = l.cur + lds_keys;
        l.cnt = l.slot + lds_keys;
        l.bas = l.cnt + lds_keys;
        l.region = l.bas + lds_keys;
        l.fa = l.region + lds_keys;
        for (uint32_t u = tid; u < m && u < lds_keys; u += T) l.h[u] = w.h[u];
        __syncthreads();
    }
    uint32_t n_prev = 0;
    bool prev_small = false;
    for (int e = 0; n_prev < m; e++) {
        const uint64_t B = ref_bucket_step(e);
        const uint32_t n_e = (uint64_t) m < B ? m : (uint32_t) B;
        const bool small = lds != nullptr && B <= lds_keys;
        if (prev_small && !small) {   // the positions so far move to where the large epochs keep them
            for (uint32_t u = tid; u < n_prev; u += T) w.cur[u] = l.cur[u];
            __syncthreads();
        }
        const OrderScratch v = small ? l : w;
        const uint32_t t32 = B < 65536ull ? (uint32_t) (0x100000000ull % B) : 0u;
        constexpr int NK = KPT > 0 ? KPT : 1;
        uint32_t hb[NK];   // (KPT > 0) the buckets of this thread's keys tid, tid + T, ...
        if constexpr (KPT > 0) {
#pragma unroll
            for (int j = 0; j < KPT; j++) {
                const uint32_t u = tid + (uint32_t) j * T;
                hb[j] = u < n_e ? ref_bucket_of(v.h[u], B, t32) : 0u;
            }
        }
        // the epoch's four passes over the keys: fn(u, bucket of u)
        auto for_keys = [&](auto fn) {
            if constexpr (KPT > 0) {
#pragma unroll
                for (int j = 0; j < KPT; j++) {
                    const uint32_t u = tid + (uint32_t) j * T;
                    if (u < n_e) fn(u, hb[j]);
                }
            } else {
                for (uint32_t u = tid; u < n_e; u += T) fn(u, ref_bucket_of(v.h[u], B, t32));
            }
        };
        for (uint64_t b = tid; b < B; b += T) v.fa[b] = 0xFFFFFFFFu;
        for (uint32_t q = tid; q < n_e; q += T) v.cnt[q] = 0;
        __syncthreads();
        // first sequence position of every bucket
        for_keys([&](uint32_t u, uint32_t bk) {
            const uint32_t q = u < n_prev ? v.cur[u] : u;
            atomicMin(&v.fa[bk], q);
        });
        __syncthreads();
        // members per bucket, counted at the bucket's first position; the arrival number is the member's slot
        for_keys([&](uint32_t u, uint32_t bk) {
            const uint32_t f = __hip_atomic_load(&v.fa[bk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v.slot[u] = atomicAdd(&v.cnt[f], 1u);
        });
        __syncthreads();
        // exclusive scan of the counts over the sequence positions: where a bucket's run starts in the sorted sequence
        {
            const uint32_t per = (n_e + T - 1) / T, q0 = tid * per;
            uint32_t sum = 0;
            for (uint32_t q = q0; q < q0 + per && q < n_e; q++) sum += __hip_atomic_load(&v.cnt[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t tot;
            uint32_t ex = block_exscan_u32<T>(sum, red, &tot);
            for (uint32_t q = q0; q < q0 + per && q < n_e; q++) {
                v.bas[q] = ex;
                ex += __hip_atomic_load(&v.cnt[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
        for_keys([&](uint32_t u, uint32_t bk) {
            const uint32_t q = u < n_prev ? v.cur[u] : u;
            const uint32_t f = __hip_atomic_load(&v.fa[bk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v.region[v.bas[f] + v.slot[u]] = q;
        });
        __syncthreads();
        for_keys([&](uint32_t u, uint32_t bk) {
            const uint32_t q = u < n_prev ? v.cur[u] : u;
            const uint32_t f = __hip_atomic_load(&v.fa[bk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t b0 = v.bas[f], c = __hip_atomic_load(&v.cnt[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t within = 0;
            for (uint32_t i = 0; i < c; i++) within += (v.region[b0 + i] < q) ? 1u : 0u;
            v.cur[u] = n_e - 1u - (b0 + within);   // (key u is always handled by the same thread: no hazard on cur)
        });
        __syncthreads();
        n_prev = n_e;
        prev_small = small;
        if ((uint64_t) m <= B) break;
    }
    if (prev_small) {   // a set that never left the small epochs
        for (uint32_t u = tid; u < m; u += T) w.cur[u] = l.cur[u];
        __syncthreads();
    }
}

}  // namespace ecal
