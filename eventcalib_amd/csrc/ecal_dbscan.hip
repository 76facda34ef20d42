// DBSCAN kernels (one workgroup per segment) and their launchers.  gfx950 only.
// Build with -ffp-contract=off: the ball predicate must be the reference's mul/add sequence.
#include "ecal_ctx.hpp"
#include "dbscan_device.hpp"
#include "dbscan_pixel.hpp"

#pragma clang fp contract(off)

namespace ecal {

// ---- LDS carve for an LDS tier of capacity CAP (all offsets multiples of 16) ----
template <int CAP>
struct TierLayout {
    static constexpr size_t c_off = 0;
    static constexpr size_t slot_off = c_off + sizeof(double) * 2 * CAP;
    static constexpr size_t anc_off = slot_off + sizeof(uint32_t) * (2 * CAP + 8);
    static constexpr size_t cur_off = anc_off + sizeof(uint16_t) * 4 * CAP;
    static constexpr size_t inv_off = cur_off + sizeof(uint16_t) * CAP;
    static constexpr size_t red_off = inv_off + sizeof(uint16_t) * CAP;
    static constexpr size_t edges_off = red_off + sizeof(uint32_t) * 48;  // red: 32 scan, 36 edge count, 40-42 any flags, 44-47 bbox
    static constexpr size_t sflags_off = edges_off + sizeof(uint32_t) * 2 * EDGE_CAP;
    static constexpr size_t bytes = sflags_off + ((CAP + 15) / 16) * 16;
    static_assert(slot_off % 16 == 0 && anc_off % 16 == 0 && cur_off % 16 == 0 && inv_off % 16 == 0 &&
                  red_off % 16 == 0 && edges_off % 16 == 0, "align");
};

template <int V>
struct Log2 {
    static constexpr uint32_t value = 1 + Log2<V / 2>::value;
};
template <>
struct Log2<1> {
    static constexpr uint32_t value = 0;
};

// bucket count of a tier = largest power of two <= CAP (label[n] + 1 + starts[nb + 2] must fit slot[2 CAP + 8])
template <int CAP>
struct NbLog {
    static constexpr uint32_t value = Log2<CAP>::value;  // Log2 rounds down
};

// Segments with lo_excl < n <= CAP are handled here; the others exit at once.
// The tier with lo_excl == 0 also writes n_clusters = 0 for empty segments.
template <int CAP, int T>
__device__ __forceinline__ void tier_segment(unsigned char *smem, uint32_t s, const double *__restrict__ xy,
                                             const uint32_t *__restrict__ seg_off,
                                             const uint32_t *__restrict__ seg_cnt, uint32_t lo_excl, double eps,
                                             uint32_t minpts, int32_t *__restrict__ labels,
                                             uint32_t *__restrict__ n_clusters) {
    const uint32_t n = seg_cnt[s];
    if (n == 0) {
        if (lo_excl == 0 && threadIdx.x == 0) n_clusters[s] = 0;
        return;
    }
    if (n <= lo_excl || n > (uint32_t) CAP) return;
    using L = TierLayout<CAP>;
    constexpr int PPT = CAP / T;
    const size_t base = seg_off[s];
    const double2 *src = reinterpret_cast<const double2 *>(xy) + base;
    uint32_t *const red = reinterpret_cast<uint32_t *>(smem + L::red_off);

    // A. load the segment; pixel-like data (integers, |v| <= 16383) takes the exact int16 path
    double2 mine[PPT];
    bool fits = true;
#pragma unroll
    for (int u = 0; u < PPT; u++) {
        const uint32_t i = threadIdx.x + u * T;
        if (i < n) {
            mine[u] = src[i];
            fits = fits && GeoI16::fits(mine[u]);
        }
    }
    if (threadIdx.x < 3) red[40 + threadIdx.x] = 0;
    __syncthreads();
    uint32_t round = 0;
    const bool int_mode = !block_any(!fits, red + 40, round);

    uint32_t total;
    if (int_mode) {
        DbWork<uint16_t, GeoI16> w;
        uint32_t *pts = reinterpret_cast<uint32_t *>(smem + L::c_off);
        w.p = pts;
        w.cs = pts;  // bucket-ordered points overwrite the pid-ordered copy once B and the count pass are done
        w.slot = reinterpret_cast<uint32_t *>(smem + L::slot_off);
        w.anc = reinterpret_cast<uint16_t *>(smem + L::anc_off);
        w.pid_s = reinterpret_cast<uint16_t *>(smem + L::cur_off);
        w.inv = reinterpret_cast<uint16_t *>(smem + L::inv_off);
        w.red = red;
        w.edges = reinterpret_cast<uint32_t *>(smem + L::edges_off);
        w.sflags = reinterpret_cast<uint8_t *>(smem + L::sflags_off);
        w.pflags = nullptr;
#pragma unroll
        for (int u = 0; u < PPT; u++)
            if (threadIdx.x + u * T < n) pts[threadIdx.x + u * T] = GeoI16::pack(mine[u]);
        __syncthreads();
        total = dbscan_segment<T, false, uint16_t, PPT, GeoI16>(w, src, n, eps, minpts, NbLog<CAP>::value, labels + base);
    } else {
        DbWork<uint16_t, GeoF64> w;
        double2 *pts = reinterpret_cast<double2 *>(smem + L::c_off);
        w.p = pts;
        w.cs = pts;
        w.slot = reinterpret_cast<uint32_t *>(smem + L::slot_off);
        w.anc = reinterpret_cast<uint16_t *>(smem + L::anc_off);
        w.pid_s = reinterpret_cast<uint16_t *>(smem + L::cur_off);
        w.inv = reinterpret_cast<uint16_t *>(smem + L::inv_off);
        w.red = red;
        w.edges = reinterpret_cast<uint32_t *>(smem + L::edges_off);
        w.sflags = reinterpret_cast<uint8_t *>(smem + L::sflags_off);
        w.pflags = nullptr;
#pragma unroll
        for (int u = 0; u < PPT; u++)
            if (threadIdx.x + u * T < n) pts[threadIdx.x + u * T] = mine[u];
        __syncthreads();
        total = dbscan_segment<T, false, uint16_t, PPT, GeoF64>(w, src, n, eps, minpts, NbLog<CAP>::value, labels + base);
    }
    if (threadIdx.x == 0) n_clusters[s] = total;
}

// todo == nullptr: workgroup b handles segment b (gridDim.x == S).  Otherwise the workgroups share the list of
// segments the pixel kernel left over (todo[0 .. *todo_count)), usually empty.
template <int CAP, int T>
__global__ __launch_bounds__(T) void dbscan_lds_kernel(const double *__restrict__ xy,
                                                       const uint32_t *__restrict__ seg_off,
                                                       const uint32_t *__restrict__ seg_cnt, uint32_t lo_excl,
                                                       double eps, uint32_t minpts, int32_t *__restrict__ labels,
                                                       uint32_t *__restrict__ n_clusters, uint32_t S,
                                                       const uint32_t *__restrict__ todo,
                                                       const uint32_t *__restrict__ todo_count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t count = todo ? *todo_count : S;
    for (uint32_t k = blockIdx.x; k < count; k += gridDim.x) {
        tier_segment<CAP, T>(smem, todo ? todo[k] : k, xy, seg_off, seg_cnt, lo_excl, eps, minpts, labels, n_clusters);
        __syncthreads();
    }
}

constexpr int BIG_T = 1024;

// Segments with n > lo_excl: workspace in global scratch, indexed by the segment's point offset.
// one segment of any size with its workspace in global scratch, indexed by the segment's point offset
__device__ __forceinline__ void big_segment(uint32_t s, uint32_t *red, uint32_t *edges, const double *__restrict__ xy,
                                            const uint32_t *__restrict__ seg_off, uint32_t n, double eps, uint32_t minpts,
                                            int32_t *__restrict__ labels, uint32_t *__restrict__ n_clusters, uint32_t *gslot, uint32_t *ganc,
                                            uint32_t *gcur, uint32_t *ginv, double2 *gcs, uint8_t *gflags) {
    const size_t base = seg_off[s];
    DbWork<uint32_t, GeoF64> w;
    const double2 *src = reinterpret_cast<const double2 *>(xy) + base;
    w.p = src;  // the caller's interleaved points are read in place
    w.cs = gcs + base;
    w.slot = gslot + 2 * base + 4 * (size_t) s;
    w.anc = ganc + 4 * base;
    w.pid_s = gcur + base;
    w.inv = ginv + base;
    w.red = red;
    w.edges = edges;
    w.sflags = gflags + 2 * base;
    w.pflags = gflags + 2 * base + n;
    // nb = largest power of two <= n/2 (>= 2048 here) so that label[n] + 1 + cursor[nb] fits slot[2n]
    uint32_t nb_log = 31u - (uint32_t) __clz((int) (n >> 1));
    if (nb_log > 20u) nb_log = 20u;
    const uint32_t total = dbscan_segment<BIG_T, true, uint32_t, 0, GeoF64>(w, src, n, eps, minpts, nb_log, labels + base);
    if (threadIdx.x == 0) n_clusters[s] = total;
}

// Segments with n > lo_excl: workspace in global scratch, indexed by the segment's point offset.
// (the segments beyond the LDS tiers, from the to-do list the pixel passes left — all S segments without one —, walked by a
// small grid: S workgroups of 1024 threads that find nothing to do cost tens of microseconds per call)
// seen (ecal_ctx::tail_seen, optional): the last kernel of the stage records what the to-do lists held (cnt_a / cnt_b)
__global__ __launch_bounds__(BIG_T) void dbscan_big_kernel(const double *__restrict__ xy,
                                                           const uint32_t *__restrict__ seg_off,
                                                           const uint32_t *__restrict__ seg_cnt, uint32_t lo_excl,
                                                           double eps, uint32_t minpts, int32_t *__restrict__ labels,
                                                           uint32_t *__restrict__ n_clusters, uint32_t *gslot,
                                                           uint32_t *ganc, uint32_t *gcur, uint32_t *ginv,
                                                           double2 *gcs, uint8_t *gflags, uint32_t S,
                                                           const uint32_t *__restrict__ todo, const uint32_t *__restrict__ todo_count,
                                                           uint32_t *seen, const uint32_t *cnt_a, const uint32_t *cnt_b) {
    __shared__ uint32_t red[48];
    __shared__ uint32_t edges[2 * EDGE_CAP];
    const uint32_t n_work = todo ? *todo_count : S;
    if (seen && blockIdx.x == 0 && threadIdx.x == 0) {
        seen[0] = cnt_a ? *cnt_a : 0u;
        seen[1] = cnt_b ? *cnt_b : 0u;
    }
    for (uint32_t kk = blockIdx.x; kk < n_work; kk += gridDim.x) {
        __syncthreads();
        const uint32_t s = todo ? todo[kk] : kk;
        const uint32_t n = seg_cnt[s];
        if (n <= lo_excl) continue;
        big_segment(s, red, edges, xy, seg_off, n, eps, minpts, labels, n_clusters, gslot, ganc, gcur, ginv, gcs, gflags);
    }
}

// The LEAN tail of the stage (ecal_ctx::tail_seen): ONE launch behind the pixel kernel that takes every listed segment — its
// doubles written first where it exists packed only, then the 4096-point LDS tier's code (1024 threads) or, beyond that, the
// global-scratch tier's.  What the tiered form does with six launches (second pixel pass, unpack, three LDS tiers, global
// scratch); the list is normally empty and this is then one ~5 us launch instead of six.  Same results: every tier computes
// the same labels (tests/test_gpu_dbscan.py compares them).
__global__ __launch_bounds__(BIG_T) void dbscan_tail_kernel(double *__restrict__ xy, const uint32_t *__restrict__ seg_off,
                                                            const uint32_t *__restrict__ seg_cnt, double eps, uint32_t minpts,
                                                            int32_t *__restrict__ labels, uint32_t *__restrict__ n_clusters,
                                                            uint32_t *gslot, uint32_t *ganc, uint32_t *gcur, uint32_t *ginv, double2 *gcs,
                                                            uint8_t *gflags, const uint32_t *__restrict__ todo,
                                                            const uint32_t *__restrict__ todo_count, const uint32_t *__restrict__ xy16,
                                                            uint32_t *__restrict__ fmt, uint32_t *seen,
                                                            const uint32_t *__restrict__ first_count /* semi: the first pass's list, todo = the second's */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using L = TierLayout<4096>;
    const uint32_t n_work = *todo_count;
    if (seen && blockIdx.x == 0 && threadIdx.x == 0) {
        seen[0] = first_count ? *first_count : n_work;
        seen[1] = first_count ? n_work : 0u;
    }
    for (uint32_t kk = blockIdx.x; kk < n_work; kk += gridDim.x) {
        __syncthreads();
        const uint32_t s = todo[kk];
        const uint32_t n = seg_cnt[s];
        if (fmt && fmt[s] == 1u) {   // packed only: the general code reads doubles (ecal_packed_points; fmt 1 -> 3)
            const uint32_t o = seg_off[s];
            double2 *out = reinterpret_cast<double2 *>(xy) + o;
            for (uint32_t i = threadIdx.x; i < n; i += BIG_T) {
                const uint32_t v = xy16[o + i];
                out[i] = make_double2((double) (int) (short) (v & 0xFFFFu), (double) (((int) v) >> 16));
            }
            __threadfence();
            __syncthreads();
            if (threadIdx.x == 0) fmt[s] = 3u;
        }
        if (n <= 4096u)
            tier_segment<4096, 1024>(smem, s, xy, seg_off, seg_cnt, 0u, eps, minpts, labels, n_clusters);
        else
            big_segment(s, reinterpret_cast<uint32_t *>(smem + L::red_off), reinterpret_cast<uint32_t *>(smem + L::edges_off), xy, seg_off, n, eps,
                        minpts, labels, n_clusters, gslot, ganc, gcur, ginv, gcs, gflags);
    }
}

}  // namespace ecal

using namespace ecal;

// general size tiers: capacity CAP points, CAP/4 threads; LDS ~36 B/point -> 4 / 2 / 1 workgroups per CU
static constexpr int CAP0 = 1024, CAP1 = 2048, CAP2 = 4096;

static int set_attrs(ecal_ctx *ctx) {
    if (ctx->attrs_set) return ECAL_OK;
    ECAL_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&dbscan_lds_kernel<CAP0, CAP0 / 4>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int) TierLayout<CAP0>::bytes));
    ECAL_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&dbscan_pixel_kernel<0, PX_CAP>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int) (PixelLayout<PX_CAP>::bytes)));
    ECAL_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&dbscan_pixel_kernel<16, PX_CAP>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int) (PixelLayout<PX_CAP>::bytes)));
    ECAL_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&dbscan_pixel_list_kernel<0, PX_CAP2>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int) PixelLayout<PX_CAP2>::bytes));
    ECAL_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&dbscan_pixel_list_kernel<16, PX_CAP2>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int) PixelLayout<PX_CAP2>::bytes));
    ECAL_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&dbscan_pixel_both_kernel<16, PX_CAP, PX_CAP2>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int) PixelLayout<PX_CAP2>::bytes));
    ECAL_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&dbscan_lds_kernel<CAP1, CAP1 / 4>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int) TierLayout<CAP1>::bytes));
    ECAL_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&dbscan_lds_kernel<CAP2, CAP2 / 4>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int) TierLayout<CAP2>::bytes));
    ECAL_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&dbscan_tail_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int) TierLayout<CAP2>::bytes));
    ctx->attrs_set = true;
    return ECAL_OK;
}

extern "C" int ecal_dbscan_batch_dev(ecal_ctx *ctx, const double *d_xy, const uint32_t *d_seg_off,
                                     const uint32_t *d_seg_cnt, uint32_t S, uint32_t n_points, uint32_t max_seg_points,
                                     double eps, uint32_t minpts, int32_t *d_labels, uint32_t *d_n_clusters,
                                     void *stream) {
    return ecal_dbscan_batch_packed_dev(ctx, const_cast<double *>(d_xy), d_seg_off, d_seg_cnt, S, n_points, max_seg_points, eps, minpts,
                                        d_labels, d_n_clusters, nullptr, stream);
}

// pk != NULL: segments marked packed are read from pk->d_xy16; the ones the pixel kernels leave to the general tiers get their
// doubles written into d_xy first (hence not const)
extern "C" int ecal_dbscan_batch_packed_dev(ecal_ctx *ctx, double *d_xy, const uint32_t *d_seg_off,
                                            const uint32_t *d_seg_cnt, uint32_t S, uint32_t n_points, uint32_t max_seg_points,
                                            double eps, uint32_t minpts, int32_t *d_labels, uint32_t *d_n_clusters,
                                            const ecal_packed_points *pk, void *stream) {
    const ecal_range range__(ctx, "ecal_dbscan_batch");
    if (!ctx) return ECAL_ERR_INVALID;
    if (pk && (!pk->d_xy16 || !pk->d_seg_fmt)) pk = nullptr;
    const uint32_t *const xy16 = pk ? pk->d_xy16 : nullptr, *const sfmt = pk ? pk->d_seg_fmt : nullptr;
    if (minpts < 1) {
        ctx->last_error = "minpts < 1 (DBSCAN::Run returns FAILED)";
        return ECAL_ERR_INVALID;
    }
    if (!(eps > 0.0) || !(eps < 1.0e300)) {
        ctx->last_error = "eps must be a positive finite number";
        return ECAL_ERR_INVALID;
    }
    if (S == 0) return ECAL_OK;
    if (!d_seg_off || !d_seg_cnt || !d_n_clusters || (n_points && (!d_xy || !d_labels))) {
        ctx->last_error = "null pointer";
        return ECAL_ERR_INVALID;
    }
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = set_attrs(ctx);
    if (rc) return rc;
    hipStream_t st = (hipStream_t) stream;
    ctx->px_tree_labels = nullptr;   // (whatever trees an earlier call left belong to ITS labels; set again below when this call exports its own)
    const uint32_t mx = max_seg_points ? max_seg_points : 0xFFFFFFFFu;
    const bool second_pass_wanted = mx > (uint32_t) PX_CAP;

    // event pixels (integer coordinates, eps < 16): the lean pixel kernel takes every segment it can and lists
    // the others; the general tiers then work that list off with a small grid (it is normally empty)
    PxGeom geom;
    const bool pixel = px_geometry(eps, &geom) && !ctx->sw.dbscan_no_pixel;
    const uint32_t *todo = nullptr, *todo_count = nullptr;
    uint32_t grid = S;
    // lean: what the pixel kernel lists (nothing, when this stage last ran) goes to ONE tail launch (dbscan_tail_kernel) instead
    // of the second pixel pass, the unpacking and the four general tiers (ecal_ctx::tail_seen); only without a size hint
    const int plan = pixel && max_seg_points == 0 ? ecal_tail_plan(ctx, ECAL_TAIL_DBSCAN) : ECAL_PLAN_TIERED;
    const bool lean = plan == ECAL_PLAN_LEAN;
    // semi: both pixel passes as always, then ONE tail launch for what the second leaves (nothing, when this stage last ran) instead
    // of the unpacking + the three LDS tiers + the global-scratch tier
    const bool semi = plan == ECAL_PLAN_SEMI && second_pass_wanted;
    const uint32_t *cnt_a = nullptr, *cnt_b = nullptr;
    if (pixel) {
        // two to-do lists: what the first pass (<= 1024 points) leaves, and what the second (<= 2048 points) leaves of that
        if ((rc = ecal_ensure(ctx, ctx->px_todo, (2 * (size_t) S + 8) * sizeof(uint32_t)))) return rc;
        uint32_t *cnt = (uint32_t *) ctx->px_todo.ptr, *list = cnt + 8, *cnt2 = cnt + 1, *list2 = list + S;
        // the lists' counters: words that are zero already, else two wiped now
        if (uint32_t *z = ecal_zero_words(ctx, st, 2)) {
            cnt = z;
            cnt2 = z + 1;
        } else {
            ECAL_HIP_TRY(ctx, hipMemsetAsync(cnt, 0, 2 * sizeof(uint32_t), st));
        }
        const uint32_t grid2 = std::min<uint32_t>(S, (uint32_t) ECAL_PX2_WG * ctx->n_cu);
        // the trees the first pass builds go out for the exact extraction's member-order kernel (ecal_ctx::px_tree): 4 B per
        // point, a flag word per segment that carries this call's number where the segment's tree is whole
        uint32_t *tree = nullptr, *tflag = nullptr;
        if (ctx->median_ties == ECAL_TIES_REFERENCE && n_points && PX_CAP <= 0xFFFE) {
            const bool fresh = ctx->px_tree_flag.cap < (size_t) S * sizeof(uint32_t);
            if (!ecal_ensure(ctx, ctx->px_tree, (size_t) n_points * sizeof(uint32_t)) &&
                !ecal_ensure(ctx, ctx->px_tree_flag, (size_t) S * sizeof(uint32_t))) {
                if (fresh) ECAL_HIP_TRY(ctx, hipMemsetAsync(ctx->px_tree_flag.ptr, 0, ctx->px_tree_flag.cap, st));   // (no stale numbers in fresh memory)
                ctx->px_tree_epoch = ctx->px_tree_epoch == 0xFFFFFFFFu ? 1u : ctx->px_tree_epoch + 1u;
                ctx->px_tree_labels = d_labels;
                ctx->px_tree_seg_off = d_seg_off;
                ctx->px_tree_S = S;
                tree = (uint32_t *) ctx->px_tree.ptr;
                tflag = (uint32_t *) ctx->px_tree_flag.ptr;
            }
        }
        const bool second_pass = second_pass_wanted && !lean;
        cnt_a = cnt;
        cnt_b = second_pass ? cnt2 : nullptr;
        // floor(eps^2) == 16 (the shipped eps = 4): the disc is compiled in; any other radius takes the generic form
        if (geom.e2i == 16 && !ctx->sw.dbscan_generic_disc) {
            if (ecal_latency_level(ctx) && second_pass)   // (few windows at work: a segment goes through the pass its size asks for in ONE launch)
                hipLaunchKernelGGL((dbscan_pixel_both_kernel<16, PX_CAP, PX_CAP2>), dim3(S), dim3(PX_T), PixelLayout<PX_CAP2>::bytes, st,
                                   d_xy, d_seg_off, d_seg_cnt, geom, minpts, d_labels, d_n_clusters, list, cnt, list2, cnt2, xy16, sfmt, tree,
                                   tflag, ctx->px_tree_epoch);
            else
            hipLaunchKernelGGL((dbscan_pixel_kernel<16, PX_CAP>), dim3(S), dim3(PX_T), PixelLayout<PX_CAP>::bytes, st,
                               d_xy, d_seg_off, d_seg_cnt, geom, minpts, d_labels, d_n_clusters, list, cnt, xy16, sfmt, tree, tflag,
                               ctx->px_tree_epoch);
            if (second_pass && !ecal_latency_level(ctx))   // (latency form: both passes were that one launch, the first pass's list is empty)
                hipLaunchKernelGGL((dbscan_pixel_list_kernel<16, PX_CAP2>), dim3(grid2), dim3(PX_T), PixelLayout<PX_CAP2>::bytes, st, d_xy,
                                   d_seg_off, d_seg_cnt, geom, minpts, d_labels, d_n_clusters, list2, cnt2,
                                   (const uint32_t *) list, (const uint32_t *) cnt, xy16, sfmt, tree, tflag, ctx->px_tree_epoch);
        } else {
            hipLaunchKernelGGL((dbscan_pixel_kernel<0, PX_CAP>), dim3(S), dim3(PX_T), PixelLayout<PX_CAP>::bytes, st,
                               d_xy, d_seg_off, d_seg_cnt, geom, minpts, d_labels, d_n_clusters, list, cnt, xy16, sfmt, tree, tflag,
                               ctx->px_tree_epoch);
            if (second_pass)
                hipLaunchKernelGGL((dbscan_pixel_list_kernel<0, PX_CAP2>), dim3(grid2), dim3(PX_T), PixelLayout<PX_CAP2>::bytes, st, d_xy,
                                   d_seg_off, d_seg_cnt, geom, minpts, d_labels, d_n_clusters, list2, cnt2,
                                   (const uint32_t *) list, (const uint32_t *) cnt, xy16, sfmt, tree, tflag, ctx->px_tree_epoch);
        }
        if (second_pass) {
            todo = list2;
            todo_count = cnt2;
        } else {
            todo = list;
            todo_count = cnt;
        }
        grid = S < 1024u ? S : 1024u;
    }
    uint32_t *const seen = pixel && ctx->tail_seen_dev ? ctx->tail_seen_dev + ECAL_TAIL_DBSCAN : nullptr;
    if (lean || semi) {
        const size_t np = n_points;
        if ((rc = ecal_ensure(ctx, ctx->big_slot, (2 * np + 4 * (size_t) S + 8) * sizeof(uint32_t)))) return rc;
        if ((rc = ecal_ensure(ctx, ctx->big_inv, np * sizeof(uint32_t)))) return rc;
        if ((rc = ecal_ensure(ctx, ctx->big_cs, np * 2 * sizeof(double)))) return rc;
        if ((rc = ecal_ensure(ctx, ctx->big_flags, 2 * np + 16))) return rc;
        if ((rc = ecal_ensure(ctx, ctx->big_anc, 4 * np * sizeof(uint32_t)))) return rc;
        if ((rc = ecal_ensure(ctx, ctx->big_cur, np * sizeof(uint32_t)))) return rc;
        hipLaunchKernelGGL(dbscan_tail_kernel, dim3(S < 256u ? S : 256u), dim3(BIG_T), TierLayout<CAP2>::bytes, st, d_xy, d_seg_off, d_seg_cnt, eps,
                           minpts, d_labels, d_n_clusters, (uint32_t *) ctx->big_slot.ptr, (uint32_t *) ctx->big_anc.ptr,
                           (uint32_t *) ctx->big_cur.ptr, (uint32_t *) ctx->big_inv.ptr, (double2 *) ctx->big_cs.ptr,
                           (uint8_t *) ctx->big_flags.ptr, todo, todo_count, pk ? pk->d_xy16 : nullptr, pk ? pk->d_seg_fmt : nullptr, seen,
                           semi ? cnt_a : (const uint32_t *) nullptr);
        ECAL_HIP_TRY(ctx, hipGetLastError());
        return ECAL_OK;
    }
    // the general tiers read doubles: what is left for them (everything, without the pixel kernels) is unpacked first
    if (pk && (rc = ecal_unpack_listed(ctx, pk, todo, todo_count, S, d_seg_off, d_seg_cnt, d_xy, 0, st))) return rc;
    hipLaunchKernelGGL((dbscan_lds_kernel<CAP0, CAP0 / 4>), dim3(grid), dim3(CAP0 / 4), TierLayout<CAP0>::bytes, st, d_xy,
                       d_seg_off, d_seg_cnt, 0u, eps, minpts, d_labels, d_n_clusters, S, todo, todo_count);
    if (mx > (uint32_t) CAP0)
        hipLaunchKernelGGL((dbscan_lds_kernel<CAP1, CAP1 / 4>), dim3(grid), dim3(CAP1 / 4), TierLayout<CAP1>::bytes, st,
                           d_xy, d_seg_off, d_seg_cnt, (uint32_t) CAP0, eps, minpts, d_labels, d_n_clusters, S, todo,
                           todo_count);
    if (mx > (uint32_t) CAP1)
        hipLaunchKernelGGL((dbscan_lds_kernel<CAP2, CAP2 / 4>), dim3(grid), dim3(CAP2 / 4), TierLayout<CAP2>::bytes, st,
                           d_xy, d_seg_off, d_seg_cnt, (uint32_t) CAP1, eps, minpts, d_labels, d_n_clusters, S, todo,
                           todo_count);
    if (mx > (uint32_t) CAP2) {
        const size_t np = n_points;
        if ((rc = ecal_ensure(ctx, ctx->big_slot, (2 * np + 4 * (size_t) S + 8) * sizeof(uint32_t)))) return rc;
        if ((rc = ecal_ensure(ctx, ctx->big_inv, np * sizeof(uint32_t)))) return rc;
        if ((rc = ecal_ensure(ctx, ctx->big_cs, np * 2 * sizeof(double)))) return rc;
        if ((rc = ecal_ensure(ctx, ctx->big_flags, 2 * np + 16))) return rc;
        if ((rc = ecal_ensure(ctx, ctx->big_anc, 4 * np * sizeof(uint32_t)))) return rc;
        if ((rc = ecal_ensure(ctx, ctx->big_cur, np * sizeof(uint32_t)))) return rc;
        hipLaunchKernelGGL(dbscan_big_kernel, dim3(S < 256u ? S : 256u), dim3(BIG_T), 0, st, d_xy, d_seg_off, d_seg_cnt, (uint32_t) CAP2,
                           eps, minpts, d_labels, d_n_clusters, (uint32_t *) ctx->big_slot.ptr, (uint32_t *) ctx->big_anc.ptr,
                           (uint32_t *) ctx->big_cur.ptr, (uint32_t *) ctx->big_inv.ptr, (double2 *) ctx->big_cs.ptr,
                           (uint8_t *) ctx->big_flags.ptr, S, todo, todo_count, seen, cnt_a, cnt_b);
    }
    ECAL_HIP_TRY(ctx, hipGetLastError());
    return ECAL_OK;
}

extern "C" int ecal_dbscan_batch(ecal_ctx *ctx, const double *xy, const uint32_t *slice_off, uint32_t S, double eps,
                                 uint32_t minpts, int32_t *labels, uint32_t *n_clusters) {
    if (!ctx) return ECAL_ERR_INVALID;
    if (minpts < 1) {
        ctx->last_error = "minpts < 1 (DBSCAN::Run returns FAILED)";
        return ECAL_ERR_INVALID;
    }
    if (!(eps > 0.0) || !(eps < 1.0e300)) {
        ctx->last_error = "eps must be a positive finite number";
        return ECAL_ERR_INVALID;
    }
    if (S == 0) return ECAL_OK;
    if (!slice_off || !n_clusters) {
        ctx->last_error = "null pointer";
        return ECAL_ERR_INVALID;
    }
    for (uint32_t s = 0; s < S; s++) {
        if (slice_off[s + 1] < slice_off[s]) {
            ctx->last_error = "slice_off must be non-decreasing";
            return ECAL_ERR_INVALID;
        }
    }
    const uint32_t base0 = slice_off[0];
    const size_t N = (size_t) slice_off[S] - base0;
    if (N && (!xy || !labels)) {
        ctx->last_error = "null pointer";
        return ECAL_ERR_INVALID;
    }
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = ecal_ensure(ctx, ctx->in_xy, (N + 1) * 2 * sizeof(double)))) return rc;
    if ((rc = ecal_ensure(ctx, ctx->in_off, (size_t) S * sizeof(uint32_t)))) return rc;
    if ((rc = ecal_ensure(ctx, ctx->in_cnt, (size_t) S * sizeof(uint32_t)))) return rc;
    if ((rc = ecal_ensure(ctx, ctx->out_labels, (N + 1) * sizeof(int32_t)))) return rc;
    if ((rc = ecal_ensure(ctx, ctx->out_ncl, (size_t) S * sizeof(uint32_t)))) return rc;
    uint32_t *h = (uint32_t *) malloc(2 * (size_t) S * sizeof(uint32_t));
    if (!h) return ECAL_ERR_NOMEM;
    uint32_t mx = 1;
    for (uint32_t s = 0; s < S; s++) {
        h[s] = slice_off[s] - base0;
        h[S + s] = slice_off[s + 1] - slice_off[s];
        if (h[S + s] > mx) mx = h[S + s];
    }
    hipStream_t st = ctx->stream;
    hipError_t e = hipSuccess;
    if (N) e = hipMemcpyAsync(ctx->in_xy.ptr, xy + 2 * (size_t) base0, N * 2 * sizeof(double), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->in_off.ptr, h, S * sizeof(uint32_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->in_cnt.ptr, h + S, S * sizeof(uint32_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);  // h is pageable: make sure it was consumed
    free(h);
    if (e != hipSuccess) {
        ctx->last_error = std::string("H2D: ") + hipGetErrorString(e);
        return ECAL_ERR_HIP;
    }
    rc = ecal_dbscan_batch_dev(ctx, (const double *) ctx->in_xy.ptr, (const uint32_t *) ctx->in_off.ptr,
                               (const uint32_t *) ctx->in_cnt.ptr, S, (uint32_t) N, mx, eps, minpts,
                               (int32_t *) ctx->out_labels.ptr, (uint32_t *) ctx->out_ncl.ptr, st);
    if (rc) return rc;
    // results land in temporaries first: caller buffers are written only on success
    int32_t *tl = N ? (int32_t *) malloc(N * sizeof(int32_t)) : nullptr;
    uint32_t *tn = (uint32_t *) malloc((size_t) S * sizeof(uint32_t));
    if ((N && !tl) || !tn) {
        free(tl);
        free(tn);
        return ECAL_ERR_NOMEM;
    }
    if (N) e = hipMemcpyAsync(tl, ctx->out_labels.ptr, N * sizeof(int32_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(tn, ctx->out_ncl.ptr, S * sizeof(uint32_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        free(tl);
        free(tn);
        ctx->last_error = std::string("dbscan/D2H: ") + hipGetErrorString(e);
        return ECAL_ERR_HIP;
    }
    if (N) memcpy(labels + base0, tl, N * sizeof(int32_t));
    memcpy(n_clusters, tn, (size_t) S * sizeof(uint32_t));
    free(tl);
    free(tn);
    return ECAL_OK;
}

#ifdef ECAL_PHASE_PROF
extern "C" int ecal_debug_phase_cycles(unsigned long long *out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(ecal::g_phase_cycles), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(ecal::g_phase_cycles), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
