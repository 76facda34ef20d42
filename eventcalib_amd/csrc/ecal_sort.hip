// Time order for a stream that arrives in any order.
//
// The reference reads the .bin file into a std::multimap<double, Event_loc_pol> (event_camera_calib/test/
// eventCameraCalib.cpp:154-163): whatever the file order, the container iterates by time stamp, and events with EQUAL
// time stamps keep their file order (multimap::emplace inserts at the upper bound of the equal range).  Every other entry
// point of this ABI takes the packed records in that order; this one produces it: a stable sort of the 25-byte records by
// their f64 time stamp — a library radix sort (rocPRIM) of (order-preserving key, record index) pairs, then one gather.
// One-off ingest work, not on the per-window path.
#include "ecal_ctx.hpp"

#include <rocprim/rocprim.hpp>

namespace ecal {

// f64 -> u64 whose unsigned order is the doubles' < order (negative values: all bits flipped; others: sign bit set)
__global__ void sort_keys_kernel(const uint8_t *__restrict__ rec, uint64_t n, uint64_t *__restrict__ keys, uint32_t *__restrict__ idx) {
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t b;
    __builtin_memcpy(&b, rec + i * 25, 8);
    if ((b << 1) == 0) b = 0;   // -0.0 and +0.0 compare equal: one key, so that their file order is kept
    keys[i] = (b >> 63) ? ~b : (b | 0x8000000000000000ull);
    idx[i] = (uint32_t) i;
}

__global__ void gather_records_kernel(const uint8_t *__restrict__ rec, const uint32_t *__restrict__ idx, uint64_t n,
                                      uint8_t *__restrict__ out) {
    // one thread per output byte: consecutive threads write consecutive bytes
    const uint64_t j = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n * 25) return;
    const uint64_t k = j / 25, o = j - k * 25;
    out[j] = rec[(uint64_t) idx[k] * 25 + o];
}

}  // namespace ecal

using namespace ecal;

extern "C" int ecal_sort_events_dev(ecal_ctx *ctx, const uint8_t *d_events, uint64_t n_events, uint8_t *d_sorted, void *stream) {
    if (!ctx || (n_events && (!d_events || !d_sorted)) || d_events == d_sorted) return ECAL_ERR_INVALID;
    if (n_events > 0xFFFFFFFFull) return ECAL_ERR_RANGE;
    if (n_events == 0) return ECAL_OK;
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t) stream;
    const size_t n = (size_t) n_events;
    size_t tmp_bytes = 0;
    uint64_t *k_in = nullptr, *k_out = nullptr;
    uint32_t *v_in = nullptr, *v_out = nullptr;
    ECAL_HIP_TRY(ctx, rocprim::radix_sort_pairs(nullptr, tmp_bytes, k_in, k_out, v_in, v_out, n, 0, 64, st));
    int rc;
    const size_t kb = (n * sizeof(uint64_t) + 255) & ~(size_t) 255, vb = (n * sizeof(uint32_t) + 255) & ~(size_t) 255;
    if ((rc = ecal_ensure(ctx, ctx->sort_scratch, 2 * kb + 2 * vb + tmp_bytes))) return rc;
    unsigned char *p = (unsigned char *) ctx->sort_scratch.ptr;
    k_in = (uint64_t *) p;
    k_out = (uint64_t *) (p + kb);
    v_in = (uint32_t *) (p + 2 * kb);
    v_out = (uint32_t *) (p + 2 * kb + vb);
    void *tmp = p + 2 * kb + 2 * vb;
    hipLaunchKernelGGL(sort_keys_kernel, dim3((uint32_t) ((n + 255) / 256)), dim3(256), 0, st, d_events, n_events, k_in, v_in);
    ECAL_HIP_TRY(ctx, rocprim::radix_sort_pairs(tmp, tmp_bytes, k_in, k_out, v_in, v_out, n, 0, 64, st));   // (stable)
    const uint64_t bytes = n_events * 25;
    hipLaunchKernelGGL(gather_records_kernel, dim3((uint32_t) ((bytes + 255) / 256)), dim3(256), 0, st, d_events, v_out, n_events, d_sorted);
    ECAL_HIP_TRY(ctx, hipGetLastError());
    return ECAL_OK;
}
