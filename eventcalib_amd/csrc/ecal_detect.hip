// Circle-candidate extraction per time-slice: DBSCAN labels of the + and - pixel sets -> filtered
// clusters -> median representatives -> mutually nearest +/- pairs -> candidate circles.
//
// Replaces CirclesEventFrame::extractFeatures between its two DBSCAN::Run calls and the
// findCirclesGrid call (event_camera_calib/src/CirclesEventFrame.cpp:89-312, fitCircle == 0 path
// :283-311) and the radius gate of :16-33.  One workgroup owns one window.
//
// Order conventions (design/04_slicing_extraction.md): a cluster's members are processed in ascending pid; its
// representative is the member of rank size/2 in the order (norm, pid) — the reference takes
// std::nth_element by norm over its BFS member order, which picks the same pixel unless two members
// tie in norm at that rank; 1-NN ties go to the smallest cluster index.
#include "ecal_ctx.hpp"
#include "block_utils.hpp"
#include "extract_window.hpp"
#include "dbscan_pixel.hpp"

#pragma clang fp contract(off)

namespace ecal {

constexpr size_t DET_LDS_BOTH = DET_LDS_BYTES + DET_TIE_INV_BYTES > DET_LDS_BYTES2 ? DET_LDS_BYTES + DET_TIE_INV_BYTES : DET_LDS_BYTES2;


// FIT = Params::fitCircle: the algebraic-fit pairing keeps two 3x4 systems in registers; compiled apart so that the default
// path (fitCircle == 0) stays below 72 VGPRs.  First pass: workgroup b takes window b.
// MODE 0: the smaller pid at tied medians; 1: the reference's pick (order given); 2: as 0 + the windows with a tied median listed
template <bool FIT, int MODE = 0>
__global__ __launch_bounds__(DET_T) void extract_kernel(
    const double *__restrict__ xy, const uint32_t *__restrict__ seg_off, const uint32_t *__restrict__ seg_cnt,
    const int32_t *__restrict__ labels, const uint32_t *__restrict__ n_clusters, DetectParams prm,
    uint32_t *__restrict__ win_info, uint32_t *__restrict__ cand_pair, double *__restrict__ cand_xyr,
    int32_t *__restrict__ kept_labels, uint32_t *__restrict__ rep, uint32_t *__restrict__ members,
    uint32_t *__restrict__ koff, uint32_t *__restrict__ ksize, uint32_t *__restrict__ sorted,
    double *__restrict__ norms, uint32_t *__restrict__ todo, uint32_t *__restrict__ todo_count, const int32_t *__restrict__ order,
    uint32_t *__restrict__ tie_list, uint32_t *__restrict__ tie_count, int32_t *__restrict__ tie_mark) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned long long red[DET_T / 64];
    __shared__ uint32_t nk_sh[4];  // kept clusters per polarity, their members per polarity
    extract_one<FIT, DET_LDS_PTS, DET_LDS_MAXC, true, false, MODE == 1, MODE == 2>(
        smem, red, nk_sh, blockIdx.x, xy, seg_off, seg_cnt, labels, n_clusters, prm, win_info, cand_pair, cand_xyr, kept_labels, rep,
        members, koff, ksize, sorted, norms, todo, todo_count, nullptr, order, tie_list, tie_count, tie_mark);
}

// The LATENCY form of the two staged passes (ecal_ctx::latency_pass: few windows hold work): workgroup b takes window b through the
// pass its size asks for — the first pass's staging (<= DET_LDS_PTS points, <= DET_LDS_MAXC clusters per polarity) or, for a window
// the first pass would LIST for the second, the second pass's (<= DET_LDS_PTS2, <= DET_LDS_MAXC2) right away; anything else is the
// first-pass code's (its global path).  Same device code per window, same results.
template <bool FIT, int MODE>
__global__ __launch_bounds__(DET_T) void extract_both_kernel(
    const double *__restrict__ xy, const uint32_t *__restrict__ seg_off, const uint32_t *__restrict__ seg_cnt,
    const int32_t *__restrict__ labels, const uint32_t *__restrict__ n_clusters, DetectParams prm,
    uint32_t *__restrict__ win_info, uint32_t *__restrict__ cand_pair, double *__restrict__ cand_xyr,
    int32_t *__restrict__ kept_labels, uint32_t *__restrict__ rep, uint32_t *__restrict__ members,
    uint32_t *__restrict__ koff, uint32_t *__restrict__ ksize, uint32_t *__restrict__ sorted,
    double *__restrict__ norms, uint32_t *__restrict__ todo, uint32_t *__restrict__ todo_count, const int32_t *__restrict__ order,
    uint32_t *__restrict__ tie_list, uint32_t *__restrict__ tie_count, int32_t *__restrict__ tie_mark) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned long long red[DET_T / 64];
    __shared__ uint32_t nk_sh[4];
    const uint32_t s = blockIdx.x;
    const uint32_t oP = seg_off[2 * s], oN = seg_off[2 * s + 1], nP = seg_cnt[2 * s], nN = seg_cnt[2 * s + 1];
    const uint32_t cP = n_clusters[2 * s], cN = n_clusters[2 * s + 1], n_all = nP + nN;
    // (extract_one's own test: what the first pass stages, and what it would hand to the second)
    const bool first_stages = oN == oP + nP && n_all <= DET_LDS_PTS && cP <= DET_LDS_MAXC && cN <= DET_LDS_MAXC;
    const bool second_stages = oN == oP + nP && n_all <= DET_LDS_PTS2 && cP <= DET_LDS_MAXC2 && cN <= DET_LDS_MAXC2;
    if (first_stages || !second_stages || nP == 0 || nN == 0)
        extract_one<FIT, DET_LDS_PTS, DET_LDS_MAXC, true, false, MODE == 1, MODE == 2>(
            smem, red, nk_sh, s, xy, seg_off, seg_cnt, labels, n_clusters, prm, win_info, cand_pair, cand_xyr, kept_labels, rep,
            members, koff, ksize, sorted, norms, todo, todo_count, nullptr, order, tie_list, tie_count, tie_mark);
    else
        extract_one<FIT, DET_LDS_PTS2, DET_LDS_MAXC2, false, false, MODE == 1, MODE == 2>(
            smem, red, nk_sh, s, xy, seg_off, seg_cnt, labels, n_clusters, prm, win_info, cand_pair, cand_xyr, kept_labels, rep,
            members, koff, ksize, sorted, norms, nullptr, nullptr, nullptr, order, tie_list, tie_count, tie_mark);
}

// the first pass over a list of windows (the exact extraction's tied windows: ecal_extract_batch_exact_dev)
template <bool FIT, int MODE = 0>
#ifndef ECAL_EFL_WAVES
#define ECAL_EFL_WAVES 6
#endif
__global__ __launch_bounds__(DET_T) __attribute__((amdgpu_waves_per_eu(ECAL_EFL_WAVES, ECAL_EFL_WAVES))) void extract_first_list_kernel(
    const double *__restrict__ xy, const uint32_t *__restrict__ seg_off, const uint32_t *__restrict__ seg_cnt,
    const int32_t *__restrict__ labels, const uint32_t *__restrict__ n_clusters, DetectParams prm,
    uint32_t *__restrict__ win_info, uint32_t *__restrict__ cand_pair, double *__restrict__ cand_xyr,
    int32_t *__restrict__ kept_labels, uint32_t *__restrict__ rep, uint32_t *__restrict__ members,
    uint32_t *__restrict__ koff, uint32_t *__restrict__ ksize, uint32_t *__restrict__ sorted,
    double *__restrict__ norms, uint32_t *__restrict__ todo, uint32_t *__restrict__ todo_count,
    const uint32_t *__restrict__ in_list, const uint32_t *__restrict__ in_count, const int32_t *__restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned long long red[DET_T / 64];
    __shared__ uint32_t nk_sh[4];
    const uint32_t count = *in_count;
    for (uint32_t k = blockIdx.x; k < count; k += gridDim.x) {
        extract_one<FIT, DET_LDS_PTS, DET_LDS_MAXC, true, false, MODE == 1, false>(
            smem, red, nk_sh, in_list[k] & 0x3FFFFFFFu /* (bits 30, 31: segments for ecal_cluster_order_list_dev to skip) */, xy, seg_off, seg_cnt, labels, n_clusters, prm, win_info, cand_pair, cand_xyr, kept_labels, rep,
            members, koff, ksize, sorted, norms, todo, todo_count, nullptr, order);
        __syncthreads();
    }
}

// second pass: the workgroups share the list of windows of DET_LDS_PTS + 1 ... DET_LDS_PTS2 points the first pass left
template <bool FIT, int MODE = 0>
__global__ __launch_bounds__(DET_T) void extract_list_kernel(
    const double *__restrict__ xy, const uint32_t *__restrict__ seg_off, const uint32_t *__restrict__ seg_cnt,
    const int32_t *__restrict__ labels, const uint32_t *__restrict__ n_clusters, DetectParams prm,
    uint32_t *__restrict__ win_info, uint32_t *__restrict__ cand_pair, double *__restrict__ cand_xyr,
    int32_t *__restrict__ kept_labels, uint32_t *__restrict__ rep, uint32_t *__restrict__ members,
    uint32_t *__restrict__ koff, uint32_t *__restrict__ ksize, uint32_t *__restrict__ sorted,
    double *__restrict__ norms, const uint32_t *__restrict__ in_list, const uint32_t *__restrict__ in_count,
    const int32_t *__restrict__ order, uint32_t *__restrict__ tie_list, uint32_t *__restrict__ tie_count, int32_t *__restrict__ tie_mark) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned long long red[DET_T / 64];
    __shared__ uint32_t nk_sh[4];
    const uint32_t count = *in_count;
    for (uint32_t k = blockIdx.x; k < count; k += gridDim.x) {
        extract_one<FIT, DET_LDS_PTS2, DET_LDS_MAXC2, false, false, MODE == 1, MODE == 2>(
            smem, red, nk_sh, in_list[k], xy, seg_off, seg_cnt, labels, n_clusters, prm, win_info, cand_pair, cand_xyr, kept_labels, rep,
            members, koff, ksize, sorted, norms, nullptr, nullptr, nullptr, order, tie_list, tie_count, tie_mark);
        __syncthreads();
    }
}


// packed points: a packed window that the staged passes will not take (more points or clusters than the second pass stages)
// reads doubles in the global path — they are written here, before the first pass.  A THREAD per window looks (256 windows per
// round of a workgroup: with a workgroup per window the look alone was a chain of 33 dependent loads per launch, 30 us of every
// pass for windows that never turn up on the shipped configuration), the workgroup unpacks the ones found.
__global__ __launch_bounds__(256) void unpack_unstaged_windows_kernel(uint32_t S, uint32_t pts_lim, uint32_t maxc_lim, const uint32_t *__restrict__ seg_off,
                                                                       const uint32_t *__restrict__ seg_cnt,
                                                                       const uint32_t *__restrict__ n_clusters,
                                                                       const uint32_t *__restrict__ xy16, uint32_t *__restrict__ fmt,
                                                                       double *__restrict__ xy) {
    __shared__ uint32_t found[256];
    __shared__ uint32_t n_found;
    for (uint32_t w0 = blockIdx.x * 256u; w0 < S; w0 += gridDim.x * 256u) {
        if (threadIdx.x == 0) n_found = 0;
        __syncthreads();
        const uint32_t w = w0 + threadIdx.x;
        if (w < S) {
            const uint2 f = *reinterpret_cast<const uint2 *>(fmt + 2 * w);
            if ((f.x & 1u) && f.x != 3u) {
                const uint2 c = *reinterpret_cast<const uint2 *>(seg_cnt + 2 * w), o = *reinterpret_cast<const uint2 *>(seg_off + 2 * w),
                            k = *reinterpret_cast<const uint2 *>(n_clusters + 2 * w);
                const bool staged = o.y == o.x + c.x && c.x + c.y <= pts_lim && k.x <= maxc_lim && k.y <= maxc_lim;
                if (!staged) found[atomicAdd(&n_found, 1u)] = w;
            }
        }
        __syncthreads();
        const uint32_t nf = n_found;
        for (uint32_t q = 0; q < nf; q++) {
            const uint32_t u = found[q];
            for (int h = 0; h < 2; h++) {
                const uint32_t o = seg_off[2 * u + h], n = seg_cnt[2 * u + h];
                double2 *out = reinterpret_cast<double2 *>(xy) + o;
                for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
                    const uint32_t v = xy16[o + i];
                    out[i] = make_double2((double) (int) (short) (v & 0xFFFFu), (double) (((int) v) >> 16));
                }
            }
        }
        __syncthreads();
        if (threadIdx.x < nf) {
            const uint32_t u = found[threadIdx.x];
            fmt[2 * u] = fmt[2 * u + 1] = 3u;
        }
        __syncthreads();
    }
}

}  // namespace ecal

using namespace ecal;

#ifdef ECAL_PHASE_PROF
extern "C" int ecal_debug_det_cycles(unsigned long long *out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(ecal::g_det_cycles), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(ecal::g_det_cycles), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

extern "C" double ecal_circle_radius_threshold(double width, double height, int rows, int cols, int asymmetric,
                                               double square_size, double circle_radius) {
    const double lo = width < height ? width : height, hi = width < height ? height : width;
    int a, b;
    if (asymmetric) {
        a = rows > 2 * cols ? rows : 2 * cols;
        b = rows < 2 * cols ? rows : 2 * cols;
    } else {
        a = rows > cols ? rows : cols;
        b = rows < cols ? rows : cols;
    }
    const double m1 = hi / a, m2 = lo / b;
    return (m1 < m2 ? m1 : m2) / square_size * circle_radius * 1.5;
}

// mode 0: plain; 1: d_order = the points' positions inside the reference's Clusters[label] (ecal_cluster_order_dev), the windows
// = in_list[0 .. *in_count) (or all S when in_list is null); 2: plain + the windows with a tied median appended to tie_list
static int extract_batch(ecal_ctx *ctx, const double *d_xy, const uint32_t *d_seg_off, const uint32_t *d_seg_cnt, const int32_t *d_labels,
                         const uint32_t *d_n_clusters, uint32_t S, uint32_t n_points, uint32_t cluster_min, uint32_t need_clusters,
                         double radius_threshold, int fit_circle, uint32_t knn_num, uint32_t *d_win_info, uint32_t *d_cand_pair,
                         double *d_cand_xyr, int32_t *d_kept_labels, uint32_t *d_rep, int mode, const int32_t *d_order,
                         const uint32_t *d_in_list, const uint32_t *d_in_count, uint32_t *d_tie_list, uint32_t *d_tie_count, int32_t *d_tie_mark,
                         void *stream, const ecal_packed_points *pk = nullptr, double tie_eps = 0.0 /* mode 2: the DBSCAN radius (0: no inline tie path) */) {
    if (!ctx) return ECAL_ERR_INVALID;
    if (S == 0) return ECAL_OK;
    if (pk && (!pk->d_xy16 || !pk->d_seg_fmt)) pk = nullptr;
    if (!d_seg_off || !d_seg_cnt || !d_n_clusters || !d_win_info ||
        (n_points && (!d_xy || !d_labels || !d_cand_pair || !d_cand_xyr || !d_kept_labels || !d_rep))) {
        ctx->last_error = "null pointer";
        return ECAL_ERR_INVALID;
    }
    if (!(radius_threshold >= 0.0)) {
        ctx->last_error = "radius_threshold must be >= 0";
        return ECAL_ERR_INVALID;
    }
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc;
    const size_t np = (size_t) n_points + 16;
    if ((rc = ecal_ensure(ctx, ctx->det_members, np * sizeof(uint32_t)))) return rc;
    if ((rc = ecal_ensure(ctx, ctx->det_koff, np * sizeof(uint32_t)))) return rc;
    if ((rc = ecal_ensure(ctx, ctx->det_ksize, np * sizeof(uint32_t)))) return rc;
    if ((rc = ecal_ensure(ctx, ctx->det_sorted, np * sizeof(uint32_t)))) return rc;
    if ((rc = ecal_ensure(ctx, ctx->det_norms, np * sizeof(double)))) return rc;
    DetectParams prm;
    prm.cluster_min = cluster_min;
    prm.need_clusters = need_clusters;
    prm.four_thr2 = 4 * radius_threshold * radius_threshold;
    prm.thr = radius_threshold;
    prm.fit_circle = fit_circle ? 1u : 0u;
    prm.knn = knn_num;
    prm.xy16 = pk ? pk->d_xy16 : nullptr;
    prm.seg_fmt = pk ? pk->d_seg_fmt : nullptr;
    if (fit_circle && (knn_num < 1 || knn_num > DET_KNN_MAX)) {
        ctx->last_error = "knn_num must be in 1..8 when fitCircle is set";
        return ECAL_ERR_INVALID;
    }
    // the exact extraction's first pass resolves the ties of small clusters itself (resolve_ties_inline) when the kd-trees of
    // these very labels and segments are at hand: the pixel DBSCAN kernel's last call on this context exported them
    if (mode == 2 && tie_eps > 0.0 && !ctx->sw.extract_no_inline_ties && ctx->px_tree_labels && ctx->px_tree_labels == (const void *) d_labels &&
        ctx->px_tree_seg_off == (const void *) d_seg_off && ctx->px_tree_S == 2 * S) {
        PxGeom geom;
        if (px_geometry(tie_eps, &geom)) {
            prm.px_tree = (const uint32_t *) ctx->px_tree.ptr;
            prm.px_tree_flag = (const uint32_t *) ctx->px_tree_flag.ptr;
            prm.px_tree_epoch = ctx->px_tree_epoch;
            prm.tie_e2i = geom.e2i;
            prm.tie_prune = geom.eps_int ? geom.Rd - 1 : geom.Rd;   // |dx| < eps for integer dx (kdtree.cpp:169)
        }
    }
    if (!ctx->det_attr_set) {
#define ECAL_DET_ATTR(K, BYTES)                                                                                        \
    ECAL_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, (int) (BYTES)))
#define ECAL_DET_ATTR3(K, BYTES)           \
    ECAL_DET_ATTR((K<false, 0>), BYTES);   \
    ECAL_DET_ATTR((K<false, 1>), BYTES);   \
    ECAL_DET_ATTR((K<false, 2>), BYTES);   \
    ECAL_DET_ATTR((K<true, 0>), BYTES);    \
    ECAL_DET_ATTR((K<true, 1>), BYTES);    \
    ECAL_DET_ATTR((K<true, 2>), BYTES)
        ECAL_DET_ATTR3(extract_kernel, DET_LDS_BYTES + DET_TIE_INV_BYTES);
        ECAL_DET_ATTR3(extract_list_kernel, DET_LDS_BYTES2);
        ECAL_DET_ATTR((extract_both_kernel<false, 2>), DET_LDS_BOTH);
        ECAL_DET_ATTR((extract_first_list_kernel<false, 1>), DET_LDS_BYTES);
        ECAL_DET_ATTR((extract_first_list_kernel<true, 1>), DET_LDS_BYTES);
#undef ECAL_DET_ATTR3
#undef ECAL_DET_ATTR
        ctx->det_attr_set = true;
    }
    // windows too large for the first pass's LDS staging but not for the second's are listed by the first pass
    if ((rc = ecal_ensure(ctx, ctx->det_todo, ((size_t) S + 4) * sizeof(uint32_t)))) return rc;
    uint32_t *cnt = (uint32_t *) ctx->det_todo.ptr, *list = cnt + 4;
    const bool second = true;
    hipStream_t st = (hipStream_t) stream;
    // the list's counter: a word that is zero already, else one wiped now
    if (uint32_t *z = ecal_zero_words(ctx, st, 1)) cnt = z;
    else ECAL_HIP_TRY(ctx, hipMemsetAsync(cnt, 0, sizeof(uint32_t), st));
    uint32_t *mem = (uint32_t *) ctx->det_members.ptr, *ko = (uint32_t *) ctx->det_koff.ptr, *ks = (uint32_t *) ctx->det_ksize.ptr,
             *so = (uint32_t *) ctx->det_sorted.ptr;
    double *no = (double *) ctx->det_norms.ptr;
    const uint32_t grid2 = S < 768u ? S : 768u;
    // the first pass over a LIST of windows (the tied ones: possibly all of them) shares the list among as many workgroups as
    // the device holds at once — six per CU
    const uint32_t gridf = S < 6u * ctx->n_cu ? S : 6u * ctx->n_cu;
#define ECAL_DET_FIRST(FIT_, MODE_)                                                                                                  \
    hipLaunchKernelGGL((extract_kernel<FIT_, MODE_>), dim3(S), dim3(DET_T), DET_LDS_BYTES + (MODE_ == 2 ? DET_TIE_INV_BYTES : 0), st, d_xy, d_seg_off, d_seg_cnt, d_labels,    \
                       d_n_clusters, prm, d_win_info, d_cand_pair, d_cand_xyr, d_kept_labels, d_rep, mem, ko, ks, so, no,              \
                       second ? list : nullptr, cnt, d_order, d_tie_list, d_tie_count, d_tie_mark)
#define ECAL_DET_FIRST_LIST(FIT_, MODE_, LIST_, COUNT_)                                                                              \
    hipLaunchKernelGGL((extract_first_list_kernel<FIT_, MODE_>), dim3(gridf), dim3(DET_T), DET_LDS_BYTES, st, d_xy, d_seg_off,         \
                       d_seg_cnt, d_labels, d_n_clusters, prm, d_win_info, d_cand_pair, d_cand_xyr, d_kept_labels, d_rep, mem, ko, ks, \
                       so, no, second ? list : nullptr, cnt, LIST_, COUNT_, d_order)
#define ECAL_DET_SECOND(FIT_, MODE_)                                                                                                 \
    hipLaunchKernelGGL((extract_list_kernel<FIT_, MODE_>), dim3(grid2), dim3(DET_T), DET_LDS_BYTES2, st, d_xy, d_seg_off, d_seg_cnt,    \
                       d_labels, d_n_clusters, prm, d_win_info, d_cand_pair, d_cand_xyr, d_kept_labels, d_rep, mem, ko, ks, so, no,    \
                       (const uint32_t *) list, (const uint32_t *) cnt, d_order, d_tie_list, d_tie_count, d_tie_mark)
    const bool fit = prm.fit_circle != 0;
    if (pk && mode != 1)   // (mode 1 follows a mode-2 pass over the same windows: done there)
        hipLaunchKernelGGL(unpack_unstaged_windows_kernel, dim3((S + 255u) / 256u < 1024u ? (S + 255u) / 256u : 1024u), dim3(256), 0, st, S, second ? DET_LDS_PTS2 : DET_LDS_PTS,
                           second ? DET_LDS_MAXC2 : DET_LDS_MAXC, d_seg_off, d_seg_cnt, d_n_clusters, (const uint32_t *) pk->d_xy16,
                           pk->d_seg_fmt, const_cast<double *>(d_xy));
    if (mode == 1 && d_in_list) {
        if (fit) ECAL_DET_FIRST_LIST(true, 1, d_in_list, d_in_count); else ECAL_DET_FIRST_LIST(false, 1, d_in_list, d_in_count);
    } else if (mode == 1) {
        if (fit) ECAL_DET_FIRST(true, 1); else ECAL_DET_FIRST(false, 1);
    } else if (mode == 2) {
        if (fit) ECAL_DET_FIRST(true, 2);
        else if (ecal_latency_level(ctx))   // (few windows at work: a window goes through the pass its size asks for in ONE launch)
            hipLaunchKernelGGL((extract_both_kernel<false, 2>), dim3(S), dim3(DET_T), DET_LDS_BOTH, st, d_xy, d_seg_off, d_seg_cnt, d_labels,
                               d_n_clusters, prm, d_win_info, d_cand_pair, d_cand_xyr, d_kept_labels, d_rep, mem, ko, ks, so, no, list, cnt,
                               d_order, d_tie_list, d_tie_count, d_tie_mark);
        else ECAL_DET_FIRST(false, 2);
    } else {
        if (fit) ECAL_DET_FIRST(true, 0); else ECAL_DET_FIRST(false, 0);
    }
    if (second) {
        if (mode == 1) {
            if (fit) ECAL_DET_SECOND(true, 1); else ECAL_DET_SECOND(false, 1);
        } else if (mode == 2) {
            // (latency form: extract_both_kernel took the second pass's windows itself and lists none)
            if (fit) ECAL_DET_SECOND(true, 2); else if (!ecal_latency_level(ctx)) ECAL_DET_SECOND(false, 2);
        } else {
            if (fit) ECAL_DET_SECOND(true, 0); else ECAL_DET_SECOND(false, 0);
        }
    }
#undef ECAL_DET_FIRST
#undef ECAL_DET_FIRST_LIST
#undef ECAL_DET_SECOND
    ECAL_HIP_TRY(ctx, hipGetLastError());
    return ECAL_OK;
}

extern "C" int ecal_extract_batch_dev(ecal_ctx *ctx, const double *d_xy, const uint32_t *d_seg_off,
                                      const uint32_t *d_seg_cnt, const int32_t *d_labels,
                                      const uint32_t *d_n_clusters, uint32_t S, uint32_t n_points,
                                      uint32_t cluster_min, uint32_t need_clusters, double radius_threshold,
                                      int fit_circle, uint32_t knn_num, uint32_t *d_win_info, uint32_t *d_cand_pair, double *d_cand_xyr,
                                      int32_t *d_kept_labels, uint32_t *d_rep, void *stream) {
    return extract_batch(ctx, d_xy, d_seg_off, d_seg_cnt, d_labels, d_n_clusters, S, n_points, cluster_min, need_clusters, radius_threshold,
                         fit_circle, knn_num, d_win_info, d_cand_pair, d_cand_xyr, d_kept_labels, d_rep, 0, nullptr, nullptr, nullptr, nullptr,
                         nullptr, nullptr, stream);
}

extern "C" int ecal_extract_batch_ordered_dev(ecal_ctx *ctx, const double *d_xy, const uint32_t *d_seg_off, const uint32_t *d_seg_cnt,
                                              const int32_t *d_labels, const uint32_t *d_n_clusters, const int32_t *d_cluster_order,
                                              uint32_t S, uint32_t n_points, uint32_t cluster_min, uint32_t need_clusters,
                                              double radius_threshold, int fit_circle, uint32_t knn_num, uint32_t *d_win_info,
                                              uint32_t *d_cand_pair, double *d_cand_xyr, int32_t *d_kept_labels, uint32_t *d_rep,
                                              void *stream) {
    if (ctx && n_points && !d_cluster_order) {
        ctx->last_error = "null pointer";
        return ECAL_ERR_INVALID;
    }
    return extract_batch(ctx, d_xy, d_seg_off, d_seg_cnt, d_labels, d_n_clusters, S, n_points, cluster_min, need_clusters, radius_threshold,
                         fit_circle, knn_num, d_win_info, d_cand_pair, d_cand_xyr, d_kept_labels, d_rep, 1, d_cluster_order, nullptr, nullptr,
                         nullptr, nullptr, nullptr, stream);
}

// The exact extraction in one call: the plain pass lists the windows in which some kept cluster's median is tied in norm (a third of
// them on the benchmark stream), ecal_cluster_order_dev works out the reference's member order for the tied clusters of those
// windows only, and the listed windows are extracted again with it.
extern "C" int ecal_cluster_order_list_dev(ecal_ctx *ctx, const double *d_xy, const uint32_t *d_seg_off, const uint32_t *d_seg_cnt,
                                           uint32_t S, double eps, const int32_t *d_labels, const uint32_t *d_n_clusters, int32_t *d_order,
                                           uint32_t *d_status, int only_tied_medians, const uint32_t *d_win_list,
                                           const uint32_t *d_win_count, void *stream);
static int extract_exact(ecal_ctx *ctx, const double *d_xy, const uint32_t *d_seg_off, const uint32_t *d_seg_cnt, const int32_t *d_labels,
                         const uint32_t *d_n_clusters, uint32_t S, uint32_t n_points, double eps, uint32_t cluster_min, uint32_t need_clusters,
                         double radius_threshold, int fit_circle, uint32_t knn_num, uint32_t *d_win_info, uint32_t *d_cand_pair,
                         double *d_cand_xyr, int32_t *d_kept_labels, uint32_t *d_rep, const ecal_packed_points *pk, void *stream);
extern "C" int ecal_extract_batch_exact_dev(ecal_ctx *ctx, const double *d_xy, const uint32_t *d_seg_off, const uint32_t *d_seg_cnt,
                                            const int32_t *d_labels, const uint32_t *d_n_clusters, uint32_t S, uint32_t n_points, double eps,
                                            uint32_t cluster_min, uint32_t need_clusters, double radius_threshold, int fit_circle,
                                            uint32_t knn_num, uint32_t *d_win_info, uint32_t *d_cand_pair, double *d_cand_xyr,
                                            int32_t *d_kept_labels, uint32_t *d_rep, void *stream) {
    return extract_exact(ctx, d_xy, d_seg_off, d_seg_cnt, d_labels, d_n_clusters, S, n_points, eps, cluster_min, need_clusters, radius_threshold,
                         fit_circle, knn_num, d_win_info, d_cand_pair, d_cand_xyr, d_kept_labels, d_rep, nullptr, stream);
}
static int extract_exact(ecal_ctx *ctx, const double *d_xy, const uint32_t *d_seg_off, const uint32_t *d_seg_cnt, const int32_t *d_labels,
                         const uint32_t *d_n_clusters, uint32_t S, uint32_t n_points, double eps, uint32_t cluster_min, uint32_t need_clusters,
                         double radius_threshold, int fit_circle, uint32_t knn_num, uint32_t *d_win_info, uint32_t *d_cand_pair,
                         double *d_cand_xyr, int32_t *d_kept_labels, uint32_t *d_rep, const ecal_packed_points *pk, void *stream) {
    if (!ctx) return ECAL_ERR_INVALID;
    if (S == 0) return ECAL_OK;
    int rc;
    if ((rc = ecal_ensure(ctx, ctx->tie_list, ((size_t) S + 4) * sizeof(uint32_t)))) return rc;
    const void *const order_was = ctx->tie_order.ptr;
    if ((rc = ecal_ensure(ctx, ctx->tie_order, ((size_t) n_points + 16) * sizeof(int32_t) + 2 * (size_t) S * sizeof(uint32_t)))) return rc;
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    // the marks below are read back from this buffer: fresh memory must not hold one by accident (every later pass leaves
    // the listed windows' slots holding ranks or -1 / -2, never the mark)
    if (ctx->tie_order.ptr != order_was)
        ECAL_HIP_TRY(ctx, hipMemsetAsync(ctx->tie_order.ptr, 0, ((size_t) n_points + 16) * sizeof(int32_t), (hipStream_t) stream));
    uint32_t *tcnt = (uint32_t *) ctx->tie_list.ptr, *tlist = tcnt + 4;
    uint32_t *const tcnt_zero = ecal_zero_words(ctx, (hipStream_t) stream, 1);
    if (tcnt_zero) tcnt = tcnt_zero;
    int32_t *order = (int32_t *) ctx->tie_order.ptr;
    uint32_t *ostatus = (uint32_t *) (order + (size_t) n_points + 16);
    ctx->tie_count_last = tcnt;   // (ecal_debug_tie_list_count)
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!tcnt_zero) ECAL_HIP_TRY(ctx, hipMemsetAsync(tcnt, 0, sizeof(uint32_t), (hipStream_t) stream));
    if ((rc = extract_batch(ctx, d_xy, d_seg_off, d_seg_cnt, d_labels, d_n_clusters, S, n_points, cluster_min, need_clusters, radius_threshold,
                            fit_circle, knn_num, d_win_info, d_cand_pair, d_cand_xyr, d_kept_labels, d_rep, 2, nullptr, nullptr, nullptr, tlist,
                            tcnt, order, stream, pk, eps)))
        return rc;
    // (only_tied_medians = 2: the tied clusters are the ones whose representative's slot the plain pass marked in `order`)
    if ((rc = ecal_cluster_order_sized(ctx, d_xy, d_seg_off, d_seg_cnt, 2 * S, n_points, eps, d_labels, d_n_clusters, order, ostatus, 2, tlist,
                                       tcnt, stream, pk))) {
        // the marks of this call must not outlive it (a later call over other windows would take them for its own)
        (void) hipMemsetAsync(order, 0, ((size_t) n_points + 16) * sizeof(int32_t), (hipStream_t) stream);
        return rc;
    }
    return extract_batch(ctx, d_xy, d_seg_off, d_seg_cnt, d_labels, d_n_clusters, S, n_points, cluster_min, need_clusters, radius_threshold,
                         fit_circle, knn_num, d_win_info, d_cand_pair, d_cand_xyr, d_kept_labels, d_rep, 1, order, tlist, tcnt, nullptr, nullptr,
                         nullptr, stream, pk);
}

// tests: the windows the last exact extraction on this context left on its list for the member-order launches (the ones its first
// pass did not resolve itself: resolve_ties_inline); synchronises `stream`
extern "C" int ecal_debug_tie_list_count(ecal_ctx *ctx, uint32_t *out, void *stream) {
    if (!ctx || !out || !ctx->tie_count_last) return ECAL_ERR_INVALID;
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(out, ctx->tie_count_last, sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t) stream));
    ECAL_HIP_TRY(ctx, hipStreamSynchronize((hipStream_t) stream));
    return ECAL_OK;
}

int ecal_extract_for_ctx(ecal_ctx *ctx, const double *d_xy, const uint32_t *d_seg_off, const uint32_t *d_seg_cnt, const int32_t *d_labels,
                         const uint32_t *d_n_clusters, uint32_t S, uint32_t n_points, double eps, uint32_t cluster_min,
                         uint32_t need_clusters, double radius_threshold, int fit_circle, uint32_t knn_num, uint32_t *d_win_info,
                         uint32_t *d_cand_pair, double *d_cand_xyr, int32_t *d_kept_labels, uint32_t *d_rep, void *stream,
                         const ecal_packed_points *pk) {
    const ecal_range range__(ctx, "ecal_extract_batch");
    if (ctx && ctx->median_ties == ECAL_TIES_REFERENCE)
        return extract_exact(ctx, d_xy, d_seg_off, d_seg_cnt, d_labels, d_n_clusters, S, n_points, eps, cluster_min, need_clusters,
                             radius_threshold, fit_circle, knn_num, d_win_info, d_cand_pair, d_cand_xyr, d_kept_labels, d_rep, pk, stream);
    return extract_batch(ctx, d_xy, d_seg_off, d_seg_cnt, d_labels, d_n_clusters, S, n_points, cluster_min, need_clusters, radius_threshold,
                         fit_circle, knn_num, d_win_info, d_cand_pair, d_cand_xyr, d_kept_labels, d_rep, 0, nullptr, nullptr, nullptr, nullptr,
                         nullptr, nullptr, stream, pk);
}

// the extraction the context's ecal_set_median_ties setting asks for, on packed points
extern "C" int ecal_extract_batch_packed_dev(ecal_ctx *ctx, double *d_xy, const uint32_t *d_seg_off, const uint32_t *d_seg_cnt,
                                             const int32_t *d_labels, const uint32_t *d_n_clusters, uint32_t S, uint32_t n_points, double eps,
                                             uint32_t cluster_min, uint32_t need_clusters, double radius_threshold, int fit_circle,
                                             uint32_t knn_num, uint32_t *d_win_info, uint32_t *d_cand_pair, double *d_cand_xyr,
                                             int32_t *d_kept_labels, uint32_t *d_rep, const ecal_packed_points *pk, void *stream) {
    return ecal_extract_for_ctx(ctx, d_xy, d_seg_off, d_seg_cnt, d_labels, d_n_clusters, S, n_points, eps, cluster_min, need_clusters,
                                radius_threshold, fit_circle, knn_num, d_win_info, d_cand_pair, d_cand_xyr, d_kept_labels, d_rep, stream, pk);
}

extern "C" int ecal_set_median_ties(ecal_ctx *ctx, int mode) {
    if (!ctx || (mode != ECAL_TIES_REFERENCE && mode != ECAL_TIES_SMALLER_PID)) return ECAL_ERR_INVALID;
    ctx->median_ties = mode;
    return ECAL_OK;
}

extern "C" int ecal_get_median_ties(const ecal_ctx *ctx) { return ctx ? ctx->median_ties : ECAL_ERR_INVALID; }
