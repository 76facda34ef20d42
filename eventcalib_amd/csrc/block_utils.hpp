// Workgroup-level primitives shared by the kernels (wave64; T = threads per workgroup).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ecal {

// two independent exclusive scans of one value per thread (packed into 64 bits), totals included.
// red: >= T/64 unsigned long long of LDS.
template <int T>
__device__ __forceinline__ void block_exscan_pair(uint32_t a, uint32_t b, unsigned long long *red, uint32_t *ea,
                                                  uint32_t *eb, uint32_t *ta, uint32_t *tb) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long v = ((unsigned long long) b << 32) | a, inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) red[wave] = inc;
    __syncthreads();
    unsigned long long pre = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < T / 64; w++) {
        const unsigned long long x = red[w];
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

// the same for two values whose totals stay below 2^16: packed into ONE 32-bit word (half the shuffles)
template <int T>
__device__ __forceinline__ void block_exscan_pair16(uint32_t a, uint32_t b, unsigned long long *red, uint32_t *ea,
                                                    uint32_t *eb, uint32_t *ta, uint32_t *tb) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t v = (b << 16) | a;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    uint32_t *r32 = reinterpret_cast<uint32_t *>(red);
    if (lane == 63) r32[wave] = inc;
    __syncthreads();
    uint32_t pre = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < T / 64; w++) {
        const uint32_t x = r32[w];
        if (w < wave) pre += x;
        tot += x;
    }
    __syncthreads();
    const uint32_t ex = pre + inc - v;
    *ea = ex & 0xFFFFu;
    *eb = ex >> 16;
    *ta = tot & 0xFFFFu;
    *tb = tot >> 16;
}

}  // namespace ecal
