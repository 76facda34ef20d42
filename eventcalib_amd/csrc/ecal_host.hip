// Host-pointer conveniences on top of the device entry points: a device-resident event stream and
// one call that runs bounds -> slicing -> DBSCAN -> candidate extraction for a batch of windows and
// copies back whatever the caller asks for.  Used by the C++ shims (host/*.hpp).
#include "ecal_ctx.hpp"

struct ecal_stream {
    ecal_ctx *ctx;
    uint8_t *d_events;
    uint64_t n_events;
};

extern "C" const uint8_t *ecal_stream_data(const ecal_stream *s) { return s ? s->d_events : nullptr; }

extern "C" int ecal_stream_create(ecal_ctx *ctx, const uint8_t *events, uint64_t n_events, ecal_stream **out) {
    if (!ctx || !out || (n_events && !events)) return ECAL_ERR_INVALID;
    *out = nullptr;
    if (n_events > 0xFFFFFFFFull) {
        ctx->last_error = "more than 2^32-1 events in one stream";
        return ECAL_ERR_RANGE;
    }
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    ecal_stream *s = new (std::nothrow) ecal_stream;
    if (!s) return ECAL_ERR_NOMEM;
    s->ctx = ctx;
    s->n_events = n_events;
    s->d_events = nullptr;
    const size_t bytes = (size_t) n_events * 25 + 16;
    hipError_t e = hipMalloc((void **) &s->d_events, bytes);
    if (e == hipSuccess && n_events)
        e = hipMemcpyAsync(s->d_events, events, (size_t) n_events * 25, hipMemcpyHostToDevice, ctx->stream);
    int *d_flag = nullptr;
    int h_flag = 0;
    if (e == hipSuccess) e = hipMalloc((void **) &d_flag, sizeof(int));
    if (e != hipSuccess) {
        ctx->last_error = std::string("ecal_stream_create: ") + hipGetErrorString(e);
        if (s->d_events) (void) hipFree(s->d_events);
        delete s;
        return e == hipErrorOutOfMemory ? ECAL_ERR_NOMEM : ECAL_ERR_HIP;
    }
    // the reference's multimap sorts by time; this ABI requires the file order to be time order
    int rc = ecal_check_sorted_dev(ctx, s->d_events, n_events, d_flag, ctx->stream);
    if (rc == ECAL_OK) {
        e = hipMemcpyAsync(&h_flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = ECAL_ERR_HIP;
    }
    (void) hipFree(d_flag);
    if (rc == ECAL_OK && h_flag) {
        ctx->last_error = "event timestamps are not non-decreasing";
        rc = ECAL_ERR_UNSORTED;
    }
    if (rc != ECAL_OK) {
        (void) hipFree(s->d_events);
        delete s;
        return rc;
    }
    *out = s;
    return ECAL_OK;
}

extern "C" void ecal_stream_destroy(ecal_stream *s) {
    if (!s) return;
    (void) hipSetDevice(s->ctx->device);
    if (s->d_events) (void) hipFree(s->d_events);
    delete s;
}

extern "C" uint64_t ecal_stream_size(const ecal_stream *s) { return s ? s->n_events : 0; }

namespace {
struct Out {
    void *host;
    ecal_devbuf *dev;
    size_t bytes;
};
}  // namespace

extern "C" int ecal_detect_batch(ecal_ctx *ctx, const ecal_stream *es, const double *t0, const double *t1, uint32_t S,
                                 const ecal_detect_params *prm, uint32_t cap_points, ecal_detect_result *res) {
    if (!ctx || !es || !prm || !res || (S && (!t0 || !t1))) return ECAL_ERR_INVALID;
    if (es->ctx != ctx) {
        ctx->last_error = "stream belongs to another context";
        return ECAL_ERR_INVALID;
    }
    if (S == 0) return ECAL_OK;
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    int rc;
    const size_t cap = (size_t) cap_points + 16;
    ecal_devbuf &b = ctx->host_pipe[0];
    (void) b;
    ecal_devbuf *B = ctx->host_pipe;  // 0 t0, 1 t1, 2 lo, 3 hi, 4 base, 5 xy, 6 seg_off, 7 seg_cnt, 8 event_point,
                                      // 9 labels, 10 ncl, 11 kept, 12 rep, 13 info, 14 pair, 15 xyr, 16 flag
    const size_t sizes[17] = {S * sizeof(double), S * sizeof(double), S * 4ul, S * 4ul, (S + 1) * 4ul,
                              cap * 16, 2ul * S * 4, 2ul * S * 4, cap * 4, cap * 4, 2ul * S * 4, cap * 4, cap * 4,
                              4ul * S * 4, cap * 8, cap * 24, 16};
    for (int i = 0; i < 17; i++)
        if ((rc = ecal_ensure(ctx, B[i], sizes[i]))) return rc;
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(B[0].ptr, t0, S * sizeof(double), hipMemcpyHostToDevice, st));
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(B[1].ptr, t1, S * sizeof(double), hipMemcpyHostToDevice, st));
    if ((rc = ecal_window_bounds_dev(ctx, es->d_events, es->n_events, (double *) B[0].ptr, (double *) B[1].ptr, S,
                                     (uint32_t *) B[2].ptr, (uint32_t *) B[3].ptr, (uint32_t *) B[4].ptr, st)))
        return rc;
    if ((rc = ecal_slice_events_dev(ctx, es->d_events, es->n_events, (uint32_t *) B[2].ptr, (uint32_t *) B[3].ptr,
                                    (uint32_t *) B[4].ptr, S, 0, cap_points, (double *) B[5].ptr,
                                    (uint32_t *) B[6].ptr, (uint32_t *) B[7].ptr, (int32_t *) B[8].ptr,
                                    (int *) B[16].ptr, st)))
        return rc;
    if ((rc = ecal_dbscan_batch_dev(ctx, (double *) B[5].ptr, (uint32_t *) B[6].ptr, (uint32_t *) B[7].ptr, 2 * S,
                                    cap_points, 0, prm->dbscan_eps, prm->dbscan_min_samples, (int32_t *) B[9].ptr,
                                    (uint32_t *) B[10].ptr, st)))
        return rc;
    if ((rc = ecal_extract_batch_dev(ctx, (double *) B[5].ptr, (uint32_t *) B[6].ptr, (uint32_t *) B[7].ptr,
                                     (int32_t *) B[9].ptr, (uint32_t *) B[10].ptr, S, cap_points, prm->cluster_min_sample,
                                     prm->need_clusters, prm->circle_radius_threshold, prm->fit_circle, prm->knn_num,
                                     (uint32_t *) B[13].ptr,
                                     (uint32_t *) B[14].ptr, (double *) B[15].ptr, (int32_t *) B[11].ptr,
                                     (uint32_t *) B[12].ptr, st)))
        return rc;
    const uint32_t M = prm->rows * prm->cols;
    if (M > 0) {
        if ((rc = ecal_ensure(ctx, ctx->host_grid_order, (size_t) S * M * sizeof(int32_t)))) return rc;
        if ((rc = ecal_ensure(ctx, ctx->host_grid_found, (size_t) S * sizeof(uint32_t)))) return rc;
        if ((rc = ecal_grid_order_dev(ctx, (uint32_t *) B[13].ptr, (uint32_t *) B[6].ptr, (double *) B[15].ptr, S,
                                      prm->rows, prm->cols, (int32_t *) ctx->host_grid_order.ptr,
                                      (uint32_t *) ctx->host_grid_found.ptr, st)))
            return rc;
        if (res->grid_order)
            ECAL_HIP_TRY(ctx, hipMemcpyAsync(res->grid_order, ctx->host_grid_order.ptr, (size_t) S * M * sizeof(int32_t),
                                             hipMemcpyDeviceToHost, st));
        if (res->grid_found)
            ECAL_HIP_TRY(ctx, hipMemcpyAsync(res->grid_found, ctx->host_grid_found.ptr, (size_t) S * sizeof(uint32_t),
                                             hipMemcpyDeviceToHost, st));
    }
    int overflow = 0;
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(&overflow, B[16].ptr, sizeof(int), hipMemcpyDeviceToHost, st));
    const Out outs[] = {{res->win_lo, &B[2], S * 4ul},        {res->win_hi, &B[3], S * 4ul},
                        {res->win_base, &B[4], (S + 1) * 4ul}, {res->xy, &B[5], (size_t) cap_points * 16},
                        {res->seg_off, &B[6], 2ul * S * 4},    {res->seg_cnt, &B[7], 2ul * S * 4},
                        {res->event_point, &B[8], (size_t) cap_points * 4},
                        {res->labels, &B[9], (size_t) cap_points * 4},
                        {res->n_clusters, &B[10], 2ul * S * 4}, {res->kept_labels, &B[11], (size_t) cap_points * 4},
                        {res->rep, &B[12], (size_t) cap_points * 4}, {res->win_info, &B[13], 4ul * S * 4},
                        {res->cand_pair, &B[14], (size_t) cap_points * 8},
                        {res->cand_xyr, &B[15], (size_t) cap_points * 24}};
    for (const Out &o : outs)
        if (o.host && o.bytes) ECAL_HIP_TRY(ctx, hipMemcpyAsync(o.host, o.dev->ptr, o.bytes, hipMemcpyDeviceToHost, st));
    ECAL_HIP_TRY(ctx, hipStreamSynchronize(st));
    if (overflow) {
        ctx->last_error = "cap_points is smaller than the number of events covered by the windows";
        return ECAL_ERR_RANGE;
    }
    return ECAL_OK;
}

// device-to-device copy on a stream (lets the Python all-reduce hook move the solver's buffer in and
// out of a torch tensor without any other HIP binding)
extern "C" int ecal_copy_dev(ecal_ctx *ctx, void *d_dst, const void *d_src, size_t bytes, void *stream, int sync) {
    if (!ctx || (bytes && (!d_dst || !d_src))) return ECAL_ERR_INVALID;
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (bytes) ECAL_HIP_TRY(ctx, hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, (hipStream_t) stream));
    if (sync) ECAL_HIP_TRY(ctx, hipStreamSynchronize((hipStream_t) stream));
    return ECAL_OK;
}

// Host-buffer form of ecal_rectify_batch_dev (what CirclesEventFrame::rectifyFeatures in host/ calls): keyframe f
// owns segments 2f (positiveEvents_) and 2f+1 (negativeEvents_) of xy / kept_labels.
extern "C" int ecal_rectify_batch(ecal_ctx *ctx, const double *xy, const uint32_t *seg_off, const uint32_t *seg_cnt,
                                  const int32_t *kept_labels, uint32_t n_points, const double *pose, uint32_t F,
                                  const double *landmarks, const ecal_rectify_params *prm, double *feat_xyr,
                                  uint32_t *feat_valid, uint32_t *frame_info) {
    if (!ctx) return ECAL_ERR_INVALID;
    if (F == 0) return ECAL_OK;
    if (!seg_off || !seg_cnt || !pose || !landmarks || !prm || !feat_xyr || !feat_valid || !frame_info ||
        (n_points && (!xy || !kept_labels))) {
        ctx->last_error = "null pointer";
        return ECAL_ERR_INVALID;
    }
    const uint32_t n = prm->rows * prm->cols;
    std::vector<uint32_t> info(4 * (size_t) F, 0), window(F);
    for (uint32_t f = 0; f < F; f++) {
        window[f] = f;
        for (int pol = 0; pol < 2; pol++) {
            const uint32_t o = seg_off[2 * f + pol], c = seg_cnt[2 * f + pol];
            if ((uint64_t) o + c > n_points) {
                ctx->last_error = "segment outside the point array";
                return ECAL_ERR_RANGE;
            }
            int32_t mx = -1;
            for (uint32_t i = 0; i < c; i++) mx = std::max(mx, kept_labels[o + i]);
            info[4 * (size_t) f + 1 + pol] = (uint32_t) (mx + 1);
        }
    }
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t np = n_points ? n_points : 1;
    const size_t bytes[11] = {np * 16, (size_t) F * 8, (size_t) F * 8, np * 4, (size_t) F * 16, (size_t) F * 4,
                              (size_t) F * 96, (size_t) n * 24, (size_t) F * n * 24, (size_t) F * n * 4, (size_t) F * 8};
    for (int i = 0; i < 11; i++)
        if (int rc = ecal_ensure(ctx, ctx->host_rect[i], bytes[i])) return rc;
    void *dp[11];
    for (int i = 0; i < 11; i++) dp[i] = ctx->host_rect[i].ptr;
    hipStream_t st = ctx->stream;
    const void *src[8] = {xy, seg_off, seg_cnt, kept_labels, info.data(), window.data(), pose, landmarks};
    for (int i = 0; i < 8; i++) {
        if ((i == 0 || i == 3) && n_points == 0) continue;
        ECAL_HIP_TRY(ctx, hipMemcpyAsync(dp[i], src[i], bytes[i], hipMemcpyHostToDevice, st));
    }
    if (int rc = ecal_rectify_batch_dev(ctx, (const double *) dp[0], (const uint32_t *) dp[1], (const uint32_t *) dp[2],
                                        (const int32_t *) dp[3], (const uint32_t *) dp[4], (const uint32_t *) dp[5],
                                        (const double *) dp[6], F, (const double *) dp[7], prm, (double *) dp[8],
                                        (uint32_t *) dp[9], (uint32_t *) dp[10], st))
        return rc;
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(feat_xyr, dp[8], bytes[8], hipMemcpyDeviceToHost, st));
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(feat_valid, dp[9], bytes[9], hipMemcpyDeviceToHost, st));
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(frame_info, dp[10], bytes[10], hipMemcpyDeviceToHost, st));
    ECAL_HIP_TRY(ctx, hipStreamSynchronize(st));
    return ECAL_OK;
}
