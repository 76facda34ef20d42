// Host-side context shared by the C-ABI translation units (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>
#include <new>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/ecal.h"

struct ecal_devbuf {
    void *ptr = nullptr;
    size_t cap = 0;
};

// Debug / test switches (ECAL_FORCE, ECAL_TRACE, ECAL_ADAPTIVE_SHAPE, …: ecal_capi.hip), read from the environment ONCE per context (ecal_init) — not per call: getenv is not safe against a
// concurrent setenv, and the entry points that consult these are the hot ones.  Tests that flip a switch on a live context call
// ecal_debug_reload_env afterwards (eventcalib_amd.capi.sync_env does it for every live context).  None of them changes a
// result, only which tier or routine produces it.
struct ecal_switches {
    // which tier / routine produces a result (parity tests run the tiers against each other)
    bool slice_no_pixel = false, dbscan_no_pixel = false, dbscan_generic_disc = false, extract_no_inline_ties = false;
    bool bounds_two_kernels = false, grid_one_wave = false, grid_serial_walk = false, solver_no_stream = false;
    int latency_forms = 0;   // ECAL_FORCE=latency_two_pass / latency_forms: 1 / 2 (ecal_latency_level)
    // traces (stderr)
    bool adaptive_trace = false, grid_debug = false, solver_trace = false, load_trace = false;
    // shape of the keyframe search's look-ahead (tests: any shape gives the same keyframes); 0 / -1: not set
    int adaptive_depth = 0, adaptive_depth_max = 0, adaptive_side = -1, adaptive_tree = -1;
    unsigned long long bo_big_arena = 0;                            // 0: not set
    double grid_tol_px = 20.0;
};
void ecal_read_switches(ecal_switches &sw);

struct ecal_ctx {
    int device = 0;
    ecal_switches sw;
    hipStream_t stream = nullptr;
    std::string last_error;
    // grow-only device scratch (never shrinks; sized for 288 GB parts: keep and reuse)
    ecal_devbuf in_xy, in_off, in_cnt, out_labels, out_ncl;  // staging for the host-pointer API
    ecal_devbuf big_slot, big_anc, big_cur, big_inv, big_cs, big_flags;    // global-scratch tier of DBSCAN
    ecal_devbuf pxs_todo;  // same for the pixel slicer
    // the insertion-order kd-trees the pixel DBSCAN kernel built (child links, 4 B per point) for the member-order kernel of the
    // exact extraction, which would otherwise build every listed segment's tree again (a third of its time): valid for the
    // labels / segment arrays of the DBSCAN call numbered px_tree_epoch, per segment when px_tree_flag[s] carries that number
    ecal_devbuf px_tree, px_tree_flag;
    uint32_t px_tree_epoch = 0, px_tree_S = 0;
    const void *px_tree_labels = nullptr, *px_tree_seg_off = nullptr;
    ecal_devbuf wb_status;  // ecal_window_bounds_dev: one word per workgroup of the look-back scan
    uint32_t wb_epoch = 0;  // ... and the number of the call that wrote it
    const uint32_t *tie_count_last = nullptr;   // the exact extraction's list counter of the last call (a zero-ring word or tie_list's own)
    int latency_pass = 0;             // the caller's word (the keyframe search): few of the launch's windows hold work — the stages take the forms that cost least LATENCY (1: a window through the tier it needs in one launch; 2, the search's tail: the slicer's third pass in that launch too)
    const int *overflow_sticky = nullptr;   // the caller's word: this overflow flag is zero and may STAY set once set — ecal_slice_events_dev does not wipe it (the keyframe search's passes: a memset between the kernels of a pass costs ~20 us with its gaps, and an overflow ends the search anyway)
    uint32_t grid_hint_windows = 0;   // ecal_grid_order_dev: windows that hold work in the next launch, by the caller's knowledge (0: the launch's size)
    hipStream_t wb_stream = nullptr;   // ... and the one stream whose calls use the table (others: the two-kernel form)
    bool wb_stream_set = false;
    hipEvent_t wb_done = nullptr;      // recorded behind every fused call of the owner: another stream takes the table over once it has fired
    ecal_devbuf px_todo;  // [4 + S] u32: count, then the segments the pixel kernel left to the general tiers
    ecal_devbuf sl_pts, sl_pol, sl_bend, sl_sorted, sl_rep, sl_pos;  // global-scratch tier of the slicer
    ecal_devbuf sort_scratch;  // ecal_sort_events_dev: keys, indices, radix-sort workspace
    ecal_devbuf bucket_tab;  // reference element order: bucket numbers per sensor pixel (ecal_events.hip)
    bool bucket_tab_built = false;
    ecal_devbuf sl_order, sl_order_big;  // reference element order: scratch of the general slicing tiers (slice_order.hpp)
    int point_order = ECAL_ORDER_REFERENCE;  // ecal_set_point_order
    int median_ties = ECAL_TIES_REFERENCE;   // ecal_set_median_ties: what the composite entry points do at tied medians
    ecal_devbuf det_members, det_koff, det_ksize, det_sorted, det_norms;  // detection stage scratch
    ecal_devbuf det_todo;  // [4 + S] u32: count, then the windows the first extraction pass left to the second
    ecal_devbuf tie_list, tie_order;  // ecal_extract_batch_exact_dev: windows with a tied median, their members' order
    ecal_devbuf bfs_host;   // staging of ecal_cluster_order
    ecal_devbuf bfs_lists;  // ecal_cluster_order_dev: neighbour lists of the range queries, one slice per workgroup
    ecal_devbuf bfs_defer;  // ecal_cluster_order_dev: the segments the first launch leaves to the later ones
    ecal_devbuf bfs_big;    // ecal_cluster_order_dev: workspace + hit-list arena of the global-scratch launch
    ecal_devbuf as_cnt, as_off;  // association: per-block counts / offsets
    ecal_devbuf as_host;         // staging of ecal_associate
    ecal_devbuf ingest_ev[2], ingest_feat;  // ecal_detect_stream_tiled: ping-pong event chunks, gathered features
    hipStream_t copy_stream = nullptr;      // uploads of the double-buffered ingest
    double *pass_pinned = nullptr;          // ecal_detect_pass: pinned window bounds in, packed verdicts out
    size_t pass_pinned_cap = 0;
    unsigned char *fetch_pinned = nullptr;  // ecal_fetch_pinned: pinned staging of the larger result downloads (a hipMemcpy into pageable memory the runtime has not seen before runs at ~0.3 GB/s: 25 - 30 ms for the keyframe search's 8 MB in the first calls of a process)
    size_t fetch_pinned_cap = 0;
    hipEvent_t ev_uploaded[2] = {nullptr, nullptr}, ev_consumed[2] = {nullptr, nullptr};
    ecal_devbuf host_pipe[17];  // staging of ecal_detect_batch
    ecal_devbuf host_grid_order, host_grid_found;
    ecal_devbuf host_rect[11];  // staging of ecal_rectify_batch
    ecal_devbuf calib_scratch;  // ecal_calibrate_views: views, blocks, reduced records
    ecal_devbuf adaptive_state, adaptive_keys, adaptive_dirs;  // ecal_detect_keyframes: per-piece window state, keyframe records, row directions per window
    hipEvent_t adaptive_ev[8] = {};             // ecal_detect_keyframes: one behind every pass in flight
    double *calib_pinned = nullptr;  // pinned host landing zone of the reduced record
    void *comm = nullptr;   // ncclComm_t (ecal_comm.hip); null = single rank
    int comm_rank = 0, comm_size = 1;
    // Words of device memory that are zero when handed out (ecal_zero_words): the to-do counters of the stage calls.  A ring
    // per stream that has asked (up to four), wiped half by half as it is used up.
    struct zero_ring {
        hipStream_t stream = nullptr;
        uint32_t *ptr = nullptr;
        uint32_t pos = 0;
        bool used = false;
    } zero_rings[4];
    // The later size tiers of a stage ("tails") normally find their to-do lists empty, and an empty launch still costs ~5 us:
    // 20 of them were 0.1 ms of every pass.  tail_seen = pinned host words into which the last kernel of a stage writes the
    // list counts it saw; when the stage's previous call saw empty lists (ECAL_TAIL_AUTO) the next call launches ONE tail
    // kernel that takes whatever is listed through the most general tier ("lean") instead of every tier in turn.  Either way
    // every listed window / segment is processed: the choice moves time, never results.
    uint32_t *tail_seen = nullptr, *tail_seen_dev = nullptr;   // [ECAL_TAIL_SLOTS]; 0xFFFFFFFF = not known yet
    int tail_mode = 0;                                         // ECAL_TAIL_AUTO / _TIERED / _LEAN (ecal_set_tail_mode)
    // a caller inside the library whose windows are second-tier ones by design (the keyframe search's adaptive windows): AUTO never
    // goes lean for it (the lean tail is the slow general kernel), but may still drop the tiers BEHIND the second (ecal_tail_plan)
    bool tail_no_lean = false;
    // roctx ranges around the stage entry points (ECAL_ROCTX=1 at ecal_init, or ecal_set_profile_ranges): the marker library
    // (librocprofiler-sdk-roctx.so) is looked up at run time — no link-time dependency —, `rocprofv3 --marker-trace
    // --kernel-trace` then shows which stage call every kernel belongs to (tracing)
    int (*roctx_push)(const char *) = nullptr;
    int (*roctx_pop)() = nullptr;
    void *roctx_lib = nullptr;
    uint32_t n_cu = 256;  // compute units of the device (grid size of the persistent kernels)
    bool attrs_set = false, slice_attrs_set = false, det_attr_set = false, bfs_attr_set = false;
    std::vector<ecal_devbuf *> all_bufs() {
        return {&in_xy, &in_off, &in_cnt, &out_labels, &out_ncl, &px_todo, &pxs_todo, &wb_status, &px_tree, &px_tree_flag, &big_slot, &big_anc, &big_cur, &big_inv, &big_cs, &big_flags,
                &sl_pts, &sl_pol, &sl_bend, &sl_sorted, &sl_rep, &sl_pos, &sl_order, &sl_order_big, &bucket_tab, &sort_scratch,
                &det_members, &det_koff, &det_ksize, &det_sorted, &det_norms, &det_todo, &bfs_lists, &bfs_defer, &bfs_big, &bfs_host, &tie_list, &tie_order, &as_cnt, &as_off,
                &host_rect[0], &host_rect[1], &host_rect[2], &host_rect[3], &host_rect[4], &host_rect[5], &host_rect[6],
                &host_rect[7], &host_rect[8], &host_rect[9], &host_rect[10],
                &host_pipe[0], &host_pipe[1], &host_pipe[2], &host_pipe[3], &host_pipe[4], &host_pipe[5],
                &host_pipe[6], &host_pipe[7], &host_pipe[8], &host_pipe[9], &host_pipe[10], &host_pipe[11],
                &host_pipe[12], &host_pipe[13], &host_pipe[14], &host_pipe[15], &host_pipe[16],
                &host_grid_order, &host_grid_found, &calib_scratch, &adaptive_state, &adaptive_keys, &adaptive_dirs, &as_host, &ingest_ev[0], &ingest_ev[1], &ingest_feat};
    }
};

#define ECAL_HIP_TRY(ctx, call)                                                                       \
    do {                                                                                              \
        hipError_t e__ = (call);                                                                      \
        if (e__ != hipSuccess) {                                                                      \
            (ctx)->last_error = std::string(#call) + ": " + hipGetErrorString(e__);                   \
            return (e__ == hipErrorOutOfMemory) ? ECAL_ERR_NOMEM : ECAL_ERR_HIP;                      \
        }                                                                                             \
    } while (0)

// packed points: the doubles of the listed segments / windows (ecal_events.hip)
int ecal_unpack_listed(ecal_ctx *ctx, const ecal_packed_points *pk, const uint32_t *d_list, const uint32_t *d_count, uint32_t S,
                       const uint32_t *d_seg_off, const uint32_t *d_seg_cnt, double *d_xy, int windows, hipStream_t st);
// ecal_cluster_order_list_dev with the callers' bound on a segment's size (ecal_bfs.hip)
int ecal_cluster_order_sized(ecal_ctx *ctx, const double *d_xy, const uint32_t *d_seg_off, const uint32_t *d_seg_cnt, uint32_t S,
                             uint32_t n_points, double eps, const int32_t *d_labels, const uint32_t *d_n_clusters, int32_t *d_order,
                             uint32_t *d_status, int only_tied_medians, const uint32_t *d_win_list, const uint32_t *d_win_count,
                             void *stream, const ecal_packed_points *pk = nullptr);
// extraction as the context's ecal_set_median_ties setting wants it (ecal_detect.hip): the exact form needs the DBSCAN radius
// ecal_grid_order_dev + the found grids' row directions (row_direction.hpp) into d_dirs[S][rows][2] (NULL: none)
int ecal_grid_order_dirs_dev(ecal_ctx *ctx, const uint32_t *d_win_info, const uint32_t *d_seg_off, const double *d_cand_xyr, uint32_t S,
                             uint32_t rows, uint32_t cols, int32_t *d_order, uint32_t *d_found, double *d_dirs, void *stream);
int ecal_extract_for_ctx(ecal_ctx *ctx, const double *d_xy, const uint32_t *d_seg_off, const uint32_t *d_seg_cnt, const int32_t *d_labels,
                         const uint32_t *d_n_clusters, uint32_t S, uint32_t n_points, double eps, uint32_t cluster_min,
                         uint32_t need_clusters, double radius_threshold, int fit_circle, uint32_t knn_num, uint32_t *d_win_info,
                         uint32_t *d_cand_pair, double *d_cand_xyr, int32_t *d_kept_labels, uint32_t *d_rep, void *stream,
                         const ecal_packed_points *pk = nullptr);
// reference element order: the per-pixel bucket table of the hot-path slicer, built on first use (ecal_events.hip)
int ecal_ensure_bucket_table(ecal_ctx *ctx, hipStream_t st);
// tail scheduling (see ecal_ctx::tail_seen): slots of the stages' lists, and "may this call run lean?"
enum { ECAL_TAIL_SLICE = 0, ECAL_TAIL_DBSCAN = 2, ECAL_TAIL_EXTRACT = 4, ECAL_TAIL_ORDER = 6, ECAL_TAIL_SLOTS = 16 };
inline bool ecal_tail_lean(const ecal_ctx *ctx, int first, int n) {
    if (ctx->tail_mode == ECAL_TAIL_LEAN) return true;
    if (ctx->tail_mode == ECAL_TAIL_TIERED || !ctx->tail_seen) return false;
    for (int k = 0; k < n; k++)
        if (__atomic_load_n(ctx->tail_seen + first + k, __ATOMIC_RELAXED) != 0u) return false;
    return true;
}
// The plan of a stage whose to-do lists sit in slots first (what the first pass listed) and first + 1 (what the second pass left
// of that): LEAN — one general launch behind the first pass; SEMI — first and second pass, then one general launch for whatever
// the second pass leaves (normally nothing: the tiers behind it are three to five launches that find their list empty); TIERED —
// every tier.  AUTO decides by what the stage's previous call on this context saw; whatever the plan, every listed window is
// processed and the results are bit-identical (tests/test_gpu_tail_modes.py).
enum { ECAL_PLAN_TIERED = 0, ECAL_PLAN_SEMI = 1, ECAL_PLAN_LEAN = 2 };
inline int ecal_tail_plan(const ecal_ctx *ctx, int first) {
    if (ctx->tail_mode == ECAL_TAIL_LEAN) return ECAL_PLAN_LEAN;
    if (ctx->tail_mode == ECAL_TAIL_TIERED || !ctx->tail_seen) return ECAL_PLAN_TIERED;
    const uint32_t a = __atomic_load_n(ctx->tail_seen + first, __ATOMIC_RELAXED), b = __atomic_load_n(ctx->tail_seen + first + 1, __ATOMIC_RELAXED);
    if (a == 0u && b == 0u && !ctx->tail_no_lean) return ECAL_PLAN_LEAN;
    if (a != 0xFFFFFFFFu && b == 0u) return ECAL_PLAN_SEMI;
    return ECAL_PLAN_TIERED;
}
// the latency forms' level of a stage call: the caller's word (ecal_ctx::latency_pass), or 1 / 2 under ECAL_FORCE=latency_two_pass /
// latency_forms (tests: the forms against the staged passes outside the keyframe search)
inline int ecal_latency_level(const ecal_ctx *ctx) { return ctx->sw.latency_forms ? ctx->sw.latency_forms : ctx->latency_pass; }
// a roctx range for the lifetime of the object (nothing when the context has no marker library loaded)
struct ecal_range {
    const ecal_ctx *c;
    ecal_range(const ecal_ctx *ctx, const char *name) : c(ctx && ctx->roctx_push ? ctx : nullptr) {
        if (c) (void) c->roctx_push(name);
    }
    ~ecal_range() {
        if (c && c->roctx_pop) (void) c->roctx_pop();
    }
    ecal_range(const ecal_range &) = delete;
    ecal_range &operator=(const ecal_range &) = delete;
};
// n <= 16 words that are zero once everything enqueued on `st` so far has run, or nullptr (more streams than rings, no memory):
// the caller then zeroes words of its own.  They stay the caller's for the next 512 calls on that stream at least.
uint32_t *ecal_zero_words(ecal_ctx *ctx, hipStream_t st, uint32_t n);
// pinned host staging of at least `bytes` (grow-only, contents NOT preserved; nullptr: no memory — the caller copies the slow way)
unsigned char *ecal_fetch_pinned(ecal_ctx *ctx, size_t bytes);
// ensure a scratch buffer of at least `bytes` (contents are NOT preserved)
int ecal_ensure(ecal_ctx *ctx, ecal_devbuf &b, size_t bytes);
