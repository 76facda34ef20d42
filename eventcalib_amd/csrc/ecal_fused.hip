// The detection pass with its three first passes in ONE kernel: a workgroup slices its window (EventFrame.cpp:10-36), runs
// DBSCAN on the window's + and - pixel sets (dbscan.h:115-265) and extracts the circle candidates
// (CirclesEventFrame.cpp:89-312) — the per-window body of the reference's worker loop (eventCameraCalib.cpp:49-56) without
// leaving the compute unit.  The stages are the device functions of the stage kernels (slice_hash.hpp, dbscan_pixel.hpp,
// extract_window.hpp), unchanged: same results, bit for bit, as ecal_slice_events_dev + ecal_dbscan_batch_dev +
// ecal_extract_batch_dev, and the same output arrays (the stages still hand points and labels over through memory — which
// the workgroup's own compute unit has just written: L2 hits instead of three cold trips to HBM per window).
//
// What the three-kernel form cannot have: at any moment a compute unit holds workgroups in DIFFERENT stages — the slicer is
// bound by vector-ALU issue, DBSCAN by chains of dependent LDS operations — so the stages overlap instead of queueing, and
// two of the three launch ramps / tails are gone.
//
// A window any stage cannot take in its first pass goes to that stage's to-do list exactly as before (plus the list of
// windows still to be extracted); the stage functions then run their second passes and general tiers over those lists.
#include "ecal_ctx.hpp"
#include "slice_hash.hpp"
#include "dbscan_pixel.hpp"
#include "extract_window.hpp"

#pragma clang fp contract(off)

namespace ecal {

struct FusedArgs {
    // slicing
    const uint8_t *rec;
    const uint32_t *win_lo, *win_hi, *win_base;
    uint32_t cap_points;
    double *xy;
    uint32_t *seg_off, *seg_cnt;
    int32_t *event_point;
    int *overflow;
    uint32_t *sl_list, *sl_cnt;
    const uint2 *bucket_tab;
    // DBSCAN
    uint32_t minpts;
    int32_t *labels;
    uint32_t *n_clusters;
    uint32_t *db_list, *db_cnt;
    // extraction
    DetectParams prm;
    uint32_t *win_info, *cand_pair;
    double *cand_xyr;
    int32_t *kept_labels;
    uint32_t *rep, *members, *koff, *ksize, *sorted;
    double *norms;
    uint32_t *det_list, *det_cnt;
    // windows whose extraction is still to be done after the stage functions' later passes
    uint32_t *def_list, *def_cnt;
};

// a value this workgroup has just written, read back: never through the scalar (constant) cache, which a neighbouring
// workgroup may have filled with the line's previous contents
__device__ __forceinline__ uint32_t reload_uniform(const uint32_t *p) {
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}

#ifndef ECAL_FUSED_WAVES
#define ECAL_FUSED_WAVES 5   // waves per SIMD the kernel is compiled for: the slicer's budget (ecal_events.hip, ECAL_RO_WAVES)
#endif

constexpr size_t fused_max(size_t a, size_t b) { return a > b ? a : b; }
constexpr size_t FUSED_SLICE_LDS = fused_max(PixHash<11>::bytes, PixHash<11>::obytes);
constexpr size_t FUSED_LDS = fused_max(fused_max(FUSED_SLICE_LDS, PixelLayout<PX_CAP>::bytes), DET_LDS_BYTES);
static_assert(PXH_T == PX_T && PX_T == DET_T, "the three stages share one workgroup");

template <int E2I>
__global__ __launch_bounds__(PXH_T) __attribute__((amdgpu_waves_per_eu(ECAL_FUSED_WAVES, ECAL_FUSED_WAVES)))
void detect_fused_kernel(const FusedArgs a, const PxGeom geom) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned long long red[DET_T / 64];
    __shared__ uint32_t nk_sh[4];
    const uint32_t w = blockIdx.x, tid = threadIdx.x;

    const bool sliced = slice_hash_window<11, true>(smem, w, a.rec, a.win_lo, a.win_hi, a.win_base, a.cap_points, a.xy, a.seg_off,
                                                    a.seg_cnt, a.event_point, a.overflow, a.sl_list, a.sl_cnt, a.bucket_tab);
    if (!sliced) {   // a later slicing pass takes the window; its segments and its extraction follow from the lists
        if (tid == 0) {
            const uint32_t at = atomicAdd(a.db_cnt, 2u);
            a.db_list[at] = 2 * w;
            a.db_list[at + 1] = 2 * w + 1;
            a.def_list[atomicAdd(a.def_cnt, 1u)] = w;
        }
        return;
    }
    __syncthreads();   // points and segment records are written (workgroup scope: one compute unit, one vector L1); LDS is free
    const uint32_t offP = reload_uniform(a.seg_off + 2 * w), offN = reload_uniform(a.seg_off + 2 * w + 1);
    const uint32_t nP = reload_uniform(a.seg_cnt + 2 * w), nN = reload_uniform(a.seg_cnt + 2 * w + 1);

    // (one copy of the DBSCAN code for both polarities: the kernel's instructions have to share the instruction cache)
    int ncP = 0, ncN = 0;
#pragma nounroll
    for (int pol = 0; pol < 2; pol++) {
        const int r = px_segment<E2I, PX_CAP, true>(smem, 2 * w + pol, a.xy, a.seg_off, a.seg_cnt, geom, a.minpts, a.labels, a.n_clusters,
                                                    a.db_list, a.db_cnt, pol ? nN : nP, pol ? offN : offP);
        if (pol) ncN = r; else ncP = r;
        __syncthreads();   // labels are written; LDS is free
    }
    if (ncP < 0 || ncN < 0) {   // a segment waits for a later DBSCAN pass: so does the window's extraction
        if (tid == 0) a.def_list[atomicAdd(a.def_cnt, 1u)] = w;
        return;
    }
    const uint32_t known[6] = {offP, offN, nP, nN, (uint32_t) ncP, (uint32_t) ncN};
    extract_one<false, DET_LDS_PTS, DET_LDS_MAXC, true, true>(smem, red, nk_sh, w, a.xy, a.seg_off, a.seg_cnt, a.labels, a.n_clusters,
                                                              a.prm, a.win_info, a.cand_pair, a.cand_xyr, a.kept_labels, a.rep,
                                                              a.members, a.koff, a.ksize, a.sorted, a.norms, a.det_list, a.det_cnt, known);
}

}  // namespace ecal

using namespace ecal;

extern "C" int ecal_detect_fused_dev(ecal_ctx *ctx, const uint8_t *d_events, uint64_t n_events, const uint32_t *d_win_lo,
                                     const uint32_t *d_win_hi, const uint32_t *d_win_base, uint32_t S, uint32_t max_win_events,
                                     uint32_t max_seg_points, uint32_t cap_points, double eps, uint32_t minpts,
                                     uint32_t cluster_min, uint32_t need_clusters, double radius_threshold, int fit_circle,
                                     uint32_t knn_num, double *d_xy, uint32_t *d_seg_off, uint32_t *d_seg_cnt,
                                     int32_t *d_event_point, int *d_overflow, int32_t *d_labels, uint32_t *d_n_clusters,
                                     uint32_t *d_win_info, uint32_t *d_cand_pair, double *d_cand_xyr, int32_t *d_kept_labels,
                                     uint32_t *d_rep, void *stream) {
    if (!ctx) return ECAL_ERR_INVALID;
    if (S == 0) return ECAL_OK;
    if ((n_events && !d_events) || !d_win_lo || !d_win_hi || !d_win_base || !d_seg_off || !d_seg_cnt || !d_overflow || !d_n_clusters ||
        !d_win_info || (cap_points && (!d_xy || !d_event_point || !d_labels || !d_cand_pair || !d_cand_xyr || !d_kept_labels || !d_rep))) {
        ctx->last_error = "null pointer";
        return ECAL_ERR_INVALID;
    }
    if (minpts < 1) {
        ctx->last_error = "minpts < 1 (DBSCAN::Run returns FAILED)";
        return ECAL_ERR_INVALID;
    }
    if (!(eps > 0.0) || !(eps < 1.0e300)) {
        ctx->last_error = "eps must be a positive finite number";
        return ECAL_ERR_INVALID;
    }
    if (!(radius_threshold >= 0.0)) {
        ctx->last_error = "radius_threshold must be >= 0";
        return ECAL_ERR_INVALID;
    }
    if (fit_circle && (knn_num < 1 || knn_num > DET_KNN_MAX)) {
        ctx->last_error = "knn_num must be in 1..8 when fitCircle is set";
        return ECAL_ERR_INVALID;
    }
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t) stream;
    int rc;
    PxGeom geom;
    // the fused kernel is compiled for the shipped configuration: reference point order, pixel DBSCAN (eps < 16),
    // fitCircle == 0 (example.yaml); any other configuration runs the same stages as three kernels
    const bool fusable = ctx->point_order == ECAL_ORDER_REFERENCE && px_geometry(eps, &geom) && !fit_circle && cap_points &&
                         !ctx->sw.no_fused_pass;
    if (fusable) {
        const size_t SS = 2 * (size_t) S;   // segments
        if ((rc = ecal_ensure(ctx, ctx->pxs_todo, (2 * (size_t) S + 8) * sizeof(uint32_t)))) return rc;
        if ((rc = ecal_ensure(ctx, ctx->px_todo, (2 * SS + 8) * sizeof(uint32_t)))) return rc;
        if ((rc = ecal_ensure(ctx, ctx->det_todo, ((size_t) S + 4) * sizeof(uint32_t)))) return rc;
        if ((rc = ecal_ensure(ctx, ctx->fused_def, ((size_t) S + 4) * sizeof(uint32_t)))) return rc;
        const size_t np = (size_t) cap_points + 16;
        if ((rc = ecal_ensure(ctx, ctx->det_members, np * sizeof(uint32_t)))) return rc;
        if ((rc = ecal_ensure(ctx, ctx->det_koff, np * sizeof(uint32_t)))) return rc;
        if ((rc = ecal_ensure(ctx, ctx->det_ksize, np * sizeof(uint32_t)))) return rc;
        if ((rc = ecal_ensure(ctx, ctx->det_sorted, np * sizeof(uint32_t)))) return rc;
        if ((rc = ecal_ensure(ctx, ctx->det_norms, np * sizeof(double)))) return rc;
        if ((rc = ecal_ensure_bucket_table(ctx, st))) return rc;
        if (!ctx->fused_attr_set) {
            ECAL_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&detect_fused_kernel<16>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int) FUSED_LDS));
            ECAL_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&detect_fused_kernel<0>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int) FUSED_LDS));
            ctx->fused_attr_set = true;
        }
        FusedArgs a;
        a.rec = d_events;
        a.win_lo = d_win_lo;
        a.win_hi = d_win_hi;
        a.win_base = d_win_base;
        a.cap_points = cap_points;
        a.xy = d_xy;
        a.seg_off = d_seg_off;
        a.seg_cnt = d_seg_cnt;
        a.event_point = d_event_point;
        a.overflow = d_overflow;
        a.sl_cnt = (uint32_t *) ctx->pxs_todo.ptr;   // the stage functions' own list layouts (ecal_events.hip, ecal_dbscan.hip, ecal_detect.hip)
        a.sl_list = a.sl_cnt + 8;
        a.bucket_tab = (const uint2 *) ctx->bucket_tab.ptr;
        a.minpts = minpts;
        a.labels = d_labels;
        a.n_clusters = d_n_clusters;
        a.db_cnt = (uint32_t *) ctx->px_todo.ptr;
        a.db_list = a.db_cnt + 8;
        a.prm.cluster_min = cluster_min;
        a.prm.need_clusters = need_clusters;
        a.prm.four_thr2 = 4 * radius_threshold * radius_threshold;
        a.prm.thr = radius_threshold;
        a.prm.fit_circle = 0u;
        a.prm.knn = knn_num;
        a.win_info = d_win_info;
        a.cand_pair = d_cand_pair;
        a.cand_xyr = d_cand_xyr;
        a.kept_labels = d_kept_labels;
        a.rep = d_rep;
        a.members = (uint32_t *) ctx->det_members.ptr;
        a.koff = (uint32_t *) ctx->det_koff.ptr;
        a.ksize = (uint32_t *) ctx->det_ksize.ptr;
        a.sorted = (uint32_t *) ctx->det_sorted.ptr;
        a.norms = (double *) ctx->det_norms.ptr;
        a.det_cnt = (uint32_t *) ctx->det_todo.ptr;
        a.det_list = ctx->sw.extract_no_second_pass ? nullptr : a.det_cnt + 4;
        a.def_cnt = (uint32_t *) ctx->fused_def.ptr;
        a.def_list = a.def_cnt + 4;
        ECAL_HIP_TRY(ctx, hipMemsetAsync(d_overflow, 0, sizeof(int), st));
        ECAL_HIP_TRY(ctx, hipMemsetAsync(a.sl_cnt, 0, 2 * sizeof(uint32_t), st));
        ECAL_HIP_TRY(ctx, hipMemsetAsync(a.db_cnt, 0, 2 * sizeof(uint32_t), st));
        ECAL_HIP_TRY(ctx, hipMemsetAsync(a.det_cnt, 0, sizeof(uint32_t), st));
        ECAL_HIP_TRY(ctx, hipMemsetAsync(a.def_cnt, 0, sizeof(uint32_t), st));
        if (geom.e2i == 16)
            hipLaunchKernelGGL(detect_fused_kernel<16>, dim3(S), dim3(PXH_T), FUSED_LDS, st, a, geom);
        else
            hipLaunchKernelGGL(detect_fused_kernel<0>, dim3(S), dim3(PXH_T), FUSED_LDS, st, a, geom);
        ECAL_HIP_TRY(ctx, hipGetLastError());
        ctx->fused_pass = true;   // the stage functions skip their first passes and go on from the lists
    }
    rc = ecal_slice_events_dev(ctx, d_events, n_events, d_win_lo, d_win_hi, d_win_base, S, max_win_events, cap_points, d_xy, d_seg_off,
                               d_seg_cnt, d_event_point, d_overflow, stream);
    if (rc == ECAL_OK)
        rc = ecal_dbscan_batch_dev(ctx, d_xy, d_seg_off, d_seg_cnt, 2 * S, cap_points, max_seg_points, eps, minpts, d_labels, d_n_clusters,
                                   stream);
    if (rc == ECAL_OK)
        rc = ecal_extract_batch_dev(ctx, d_xy, d_seg_off, d_seg_cnt, d_labels, d_n_clusters, S, cap_points, cluster_min, need_clusters,
                                    radius_threshold, fit_circle, knn_num, d_win_info, d_cand_pair, d_cand_xyr, d_kept_labels, d_rep,
                                    stream);
    ctx->fused_pass = false;
    return rc;
}
