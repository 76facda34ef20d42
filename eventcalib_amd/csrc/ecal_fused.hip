// ecal_detect_fused_dev: the body of the reference's worker loop per window in ONE call (event_camera_calib/test/
// eventCameraCalib.cpp:49-56: the EventFrame constructor, extractFeatures with its two DBSCAN::Run calls) — slicing, DBSCAN over the
// 2 S segments and candidate extraction, enqueued back to back on the caller's stream: the three stage entry points with the same
// arguments, the same output arrays, the same results.
// (Rounds 2 - 5 also carried ONE kernel per window through the three stages on one compute unit, detect_fused_kernel: measured
// 2.72 ms against 2.36 ms for the three stage kernels on the 50 M-event stream — the fused kernel holds 5 workgroups per CU through
// stages that alone run at 6 - 7 —, so it never was the timed path; profiles/experiments/r06_detect_fused_kernel.patch.)
#include "ecal_ctx.hpp"

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
    if (fit_circle && (knn_num < 1 || knn_num > 8)) {
        ctx->last_error = "knn_num must be in 1..8 when fitCircle is set";
        return ECAL_ERR_INVALID;
    }
    int rc = ecal_slice_events_dev(ctx, d_events, n_events, d_win_lo, d_win_hi, d_win_base, S, max_win_events, cap_points, d_xy, d_seg_off,
                               d_seg_cnt, d_event_point, d_overflow, stream);
    if (rc == ECAL_OK)
        rc = ecal_dbscan_batch_dev(ctx, d_xy, d_seg_off, d_seg_cnt, 2 * S, cap_points, max_seg_points, eps, minpts, d_labels, d_n_clusters,
                                   stream);
    if (rc == ECAL_OK)
        rc = ecal_extract_batch_dev(ctx, d_xy, d_seg_off, d_seg_cnt, d_labels, d_n_clusters, S, cap_points, cluster_min, need_clusters,
                                    radius_threshold, fit_circle, knn_num, d_win_info, d_cand_pair, d_cand_xyr, d_kept_labels, d_rep,
                                    stream);
    return rc;
}
