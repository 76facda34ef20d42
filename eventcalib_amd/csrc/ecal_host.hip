// Host-pointer conveniences on top of the device entry points: a device-resident event stream and
// one call that runs bounds -> slicing -> DBSCAN -> candidate extraction for a batch of windows and
// copies back whatever the caller asks for.  Used by the C++ shims (host/*.hpp).
#include <chrono>
#include <algorithm>
#include <math.h>
#include <fcntl.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <thread>
#include <atomic>
#include <mutex>
#include <condition_variable>
#include "ecal_ctx.hpp"

struct ecal_stream {
    ecal_ctx *ctx;
    uint8_t *d_events;
    uint64_t n_events;
};
static int finish_stream(ecal_ctx *ctx, ecal_stream *s);

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
    if (e != hipSuccess) {
        ctx->last_error = std::string("ecal_stream_create: ") + hipGetErrorString(e);
        if (s->d_events) (void) hipFree(s->d_events);
        delete s;
        return e == hipErrorOutOfMemory ? ECAL_ERR_NOMEM : ECAL_ERR_HIP;
    }
    const int rc = finish_stream(ctx, s);
    if (rc != ECAL_OK) return rc;
    *out = s;
    return ECAL_OK;
}

// The records are on the device: bring them into the multimap's order if they are not in it (frees the stream on an error)
static int finish_stream(ecal_ctx *ctx, ecal_stream *s) {
    const uint64_t n_events = s->n_events;
    const size_t bytes = (size_t) n_events * 25 + 16;
    int *d_flag = nullptr;
    int h_flag = 0;
    hipError_t e = hipMalloc((void **) &d_flag, sizeof(int));
    if (e != hipSuccess) {
        ctx->last_error = std::string("ecal_stream_create: ") + hipGetErrorString(e);
        (void) hipFree(s->d_events);
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
        // not in time order: bring it into the order the reference's multimap iterates in (stable by time stamp,
        // eventCameraCalib.cpp:154-163) with one device sort
        uint8_t *d_sorted = nullptr;
        e = hipMalloc((void **) &d_sorted, bytes);
        if (e != hipSuccess) {
            rc = e == hipErrorOutOfMemory ? ECAL_ERR_NOMEM : ECAL_ERR_HIP;
        } else {
            rc = ecal_sort_events_dev(ctx, s->d_events, n_events, d_sorted, ctx->stream);
            if (rc == ECAL_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = ECAL_ERR_HIP;
            if (rc == ECAL_OK) {
                (void) hipFree(s->d_events);
                s->d_events = d_sorted;
            } else {
                (void) hipFree(d_sorted);
            }
        }
    }
    if (rc != ECAL_OK) {
        (void) hipFree(s->d_events);
        delete s;
        return rc;
    }
    return ECAL_OK;
}

// A .bin file of 25-byte records (EventStream's format, Event.hpp:41-47) into a resident stream — the loop of
// eventCameraCalib.cpp:154-163 (`if (timeStamp >= StartTime) emplace`, stop at the first record with timeStamp >= EndTime when
// an end is set) without the host container: the file is read in chunks by a few threads into two pinned buffers, the
// upload of a chunk runs while the next one is read.  has_end == 0: to the end of the file.
extern "C" int ecal_stream_create_from_file(ecal_ctx *ctx, const char *path, double start_time, int has_end, double end_time,
                                            ecal_stream **out) {
    if (!ctx || !path || !out) return ECAL_ERR_INVALID;
    *out = nullptr;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) {
        ctx->last_error = std::string("ecal_stream_create_from_file: cannot open ") + path;
        return ECAL_ERR_INVALID;
    }
    struct stat sb;
    if (fstat(fd, &sb) != 0) {
        close(fd);
        return ECAL_ERR_INVALID;
    }
    const uint64_t n_file = (uint64_t) sb.st_size / 25u;
    if (n_file > 0xFFFFFFFFull) {
        close(fd);
        ctx->last_error = "more than 2^32-1 events in one stream";
        return ECAL_ERR_RANGE;
    }
    if (hipSetDevice(ctx->device) != hipSuccess) {
        close(fd);
        return ECAL_ERR_HIP;
    }
    ecal_stream *s = new (std::nothrow) ecal_stream;
    if (!s) {
        close(fd);
        return ECAL_ERR_NOMEM;
    }
    s->ctx = ctx;
    s->n_events = 0;
    s->d_events = nullptr;
    // Round 4: NT reader threads that LIVE for the whole file, each with a pinned buffer of its own and every NT-th chunk
    // (round 3 started four threads per 25 MiB chunk, all on the same chunk: 48 x 4 thread starts and never more than one
    // chunk in flight — 0.38 s for 1.25 GB in the driver's run, 3.3 GB/s).  A thread reads its chunk, applies the reading
    // rule, waits for its turn (uploads are issued in file order: the kept records are contiguous in HBM) and enqueues the
    // copy; reads of NT chunks overlap each other and the uploads.
    constexpr uint64_t CH = 640u * 1024u;                // records per chunk (16 MB)
    constexpr int NT = 6;                                // reader threads = pinned buffers
    uint8_t *pin[NT] = {};
    hipEvent_t done[NT] = {};
    const bool load_trace = ctx->sw.load_trace;   // (ECAL_TRACE=load: where the loader's time goes, on stderr)
    const auto lt0 = std::chrono::steady_clock::now();
    auto lt = [&](const char *what) {
        if (load_trace)
            fprintf(stderr, "ecal_stream_create_from_file: %-28s %.4f s\n", what,
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - lt0).count());
    };
    hipError_t e = hipMalloc((void **) &s->d_events, (size_t) n_file * 25 + 16);
    lt("hipMalloc");
    const uint64_t n_chunks = (n_file + CH - 1) / CH;
    const int nt = (int) std::min<uint64_t>(NT, std::max<uint64_t>(n_chunks, 1));
    for (int b = 0; b < nt && e == hipSuccess; b++) {
        e = hipHostMalloc((void **) &pin[b], (size_t) std::min<uint64_t>(CH, std::max<uint64_t>(n_file, 1)) * 25, hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&done[b], hipEventDisableTiming);
    }
    lt("pinned buffers + events");
    auto cleanup = [&]() {
        for (int b = 0; b < NT; b++) {
            if (done[b]) (void) hipEventDestroy(done[b]);
            if (pin[b]) (void) hipHostFree(pin[b]);
        }
        close(fd);
    };
    if (e != hipSuccess) {
        ctx->last_error = std::string("ecal_stream_create_from_file: ") + hipGetErrorString(e);
        cleanup();
        if (s->d_events) (void) hipFree(s->d_events);
        delete s;
        return e == hipErrorOutOfMemory ? ECAL_ERR_NOMEM : ECAL_ERR_HIP;
    }
    uint64_t kept = 0;
    std::atomic<bool> stop{false}, io_error{false};
    std::mutex mu;
    std::condition_variable cv;
    uint64_t turn = 0;   // the chunk whose upload is enqueued next (under mu)
    auto reader = [&](int b) {
        if (hipSetDevice(ctx->device) != hipSuccess) io_error = true;
        for (uint64_t k = (uint64_t) b; k < n_chunks; k += (uint64_t) nt) {
            const uint64_t r0 = k * CH, nr = std::min<uint64_t>(CH, n_file - r0);
            uint64_t n_out = 0;
            bool stops_here = false;
            if (!stop && !io_error) {
                if (k >= (uint64_t) nt && hipEventSynchronize(done[b]) != hipSuccess) io_error = true;   // the buffer's previous upload
                size_t want = (size_t) nr * 25, got = 0;
                while (got < want && !io_error) {
                    const ssize_t rd = pread(fd, pin[b] + got, want - got, (off_t) (r0 * 25 + got));
                    if (rd <= 0) io_error = true;
                    else got += (size_t) rd;
                }
                // the reference's loop (eventCameraCalib.cpp:154-163); a chunk whose records all pass goes up as it is
                bool plain = true;
                for (uint64_t i = 0; i < nr && !io_error; i++) {
                    double ts;
                    memcpy(&ts, pin[b] + i * 25, 8);
                    if (!(ts >= start_time) || (has_end && ts >= end_time)) {
                        plain = false;
                        break;
                    }
                }
                n_out = nr;
                if (!plain && !io_error) {
                    n_out = 0;
                    for (uint64_t i = 0; i < nr; i++) {
                        double ts;
                        memcpy(&ts, pin[b] + i * 25, 8);
                        if (has_end && ts >= end_time) {
                            stops_here = true;
                            break;
                        }
                        if (ts >= start_time) {
                            if (n_out != i) memmove(pin[b] + n_out * 25, pin[b] + i * 25, 25);
                            n_out++;
                        }
                    }
                }
            }
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&]() { return turn == k; });
            if (!stop && !io_error) {   // (a chunk behind the one that met EndTime is not part of the stream)
                if (n_out && hipMemcpyAsync(s->d_events + kept * 25, pin[b], (size_t) n_out * 25, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
                    io_error = true;
                if (hipEventRecord(done[b], ctx->stream) != hipSuccess) io_error = true;
                kept += n_out;
                if (stops_here) stop = true;
            }
            turn = k + 1;
            lk.unlock();
            cv.notify_all();
        }
    };
    {
        std::vector<std::thread> th;
        for (int b = 1; b < nt; b++) th.emplace_back(reader, b);
        if (nt > 0 && n_chunks > 0) reader(0);
        for (auto &t : th) t.join();
    }
    lt("reads done, copies enqueued");
    const bool synced = hipStreamSynchronize(ctx->stream) == hipSuccess;
    lt("copies done");
    cleanup();
    lt("buffers released");
    if (io_error || !synced) {
        ctx->last_error = std::string("ecal_stream_create_from_file: read or copy failed for ") + path;
        (void) hipFree(s->d_events);
        delete s;
        return ECAL_ERR_HIP;
    }
    s->n_events = kept;
    const int rc = finish_stream(ctx, s);
    if (rc != ECAL_OK) return rc;
    *out = s;
    return ECAL_OK;
}

// time stamps of the first and the last record of the stream (0 events: both 0)
extern "C" int ecal_stream_times(const ecal_stream *s, double *first, double *last) {
    if (!s || !first || !last) return ECAL_ERR_INVALID;
    *first = *last = 0.0;
    if (s->n_events == 0) return ECAL_OK;
    if (hipSetDevice(s->ctx->device) != hipSuccess) return ECAL_ERR_HIP;
    if (hipMemcpy(first, s->d_events, 8, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(last, s->d_events + (s->n_events - 1) * 25, 8, hipMemcpyDeviceToHost) != hipSuccess)
        return ECAL_ERR_HIP;
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
    if ((rc = ecal_extract_for_ctx(ctx, (double *) B[5].ptr, (uint32_t *) B[6].ptr, (uint32_t *) B[7].ptr,
                                     (int32_t *) B[9].ptr, (uint32_t *) B[10].ptr, S, cap_points, prm->dbscan_eps, prm->cluster_min_sample,
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

// rectifyFeatures for F keyframes given by their time windows, everything on the device (what host/event_calib_ini.hpp's
// batched path calls): EventFrame + extractFeatures up to the kept clusters for every window (the stages of ecal_detect_batch,
// nothing copied back), then ecal_rectify_batch_dev with keyframe f = window f.
extern "C" int ecal_rectify_keyframes(ecal_ctx *ctx, const ecal_stream *es, const double *durations, uint32_t F,
                                      const ecal_detect_params *dprm, const double *poses, const double *landmarks,
                                      const ecal_rectify_params *rprm, double *feat_xyr, uint32_t *feat_valid, uint32_t *frame_info) {
    if (!ctx || !es || !dprm || !rprm || (F && (!durations || !poses || !landmarks || !feat_xyr || !feat_valid || !frame_info)))
        return ECAL_ERR_INVALID;
    if (es->ctx != ctx) {
        ctx->last_error = "stream belongs to another context";
        return ECAL_ERR_INVALID;
    }
    if (F == 0) return ECAL_OK;
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    int rc;
    std::vector<double> t0(F), t1(F);
    for (uint32_t f = 0; f < F; f++) {
        t0[f] = durations[2 * f];
        t1[f] = durations[2 * f + 1];
    }
    // slots = the events the windows cover: one bounds pass tells
    ecal_devbuf *B = ctx->host_pipe;
    const size_t pre[5] = {F * sizeof(double), F * sizeof(double), F * 4ul, F * 4ul, (F + 1) * 4ul};
    for (int i = 0; i < 5; i++)
        if ((rc = ecal_ensure(ctx, B[i], pre[i]))) return rc;
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(B[0].ptr, t0.data(), F * sizeof(double), hipMemcpyHostToDevice, st));
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(B[1].ptr, t1.data(), F * sizeof(double), hipMemcpyHostToDevice, st));
    if ((rc = ecal_window_bounds_dev(ctx, es->d_events, es->n_events, (double *) B[0].ptr, (double *) B[1].ptr, F, (uint32_t *) B[2].ptr,
                                     (uint32_t *) B[3].ptr, (uint32_t *) B[4].ptr, st)))
        return rc;
    uint32_t total = 0;
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(&total, (uint32_t *) B[4].ptr + F, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    ECAL_HIP_TRY(ctx, hipStreamSynchronize(st));
    ecal_detect_params dp = *dprm;
    dp.rows = dp.cols = 0;   // (no grid ordering: rectifyFeatures works on the clusters)
    ecal_detect_result none;
    memset(&none, 0, sizeof(none));
    // every size tier at work: a keyframe's window is 4 - 10 steps long (second- and third-tier work by construction), and what
    // the stages' previous call saw — the keyframe search's last pass, with next to nothing in its lists — says nothing about
    // this one: left to the automatic plan the whole batch went through the one slow general launch (9.4 ms instead of 3)
    const int was_mode = ctx->tail_mode;
    ctx->tail_mode = ECAL_TAIL_TIERED;
    rc = ecal_detect_batch(ctx, es, t0.data(), t1.data(), F, &dp, total + 64u, &none);
    ctx->tail_mode = was_mode;
    if (rc) return rc;
    const uint32_t n = rprm->rows * rprm->cols;
    ecal_devbuf *R = ctx->host_rect;
    const size_t rs[6] = {(size_t) F * 12 * 8, (size_t) n * 3 * 8, (size_t) F * 4, (size_t) F * n * 24, (size_t) F * n * 4, (size_t) F * 8};
    for (int i = 0; i < 6; i++)
        if ((rc = ecal_ensure(ctx, R[i], rs[i]))) return rc;
    std::vector<uint32_t> ident(F);
    for (uint32_t f = 0; f < F; f++) ident[f] = f;
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(R[0].ptr, poses, rs[0], hipMemcpyHostToDevice, st));
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(R[1].ptr, landmarks, rs[1], hipMemcpyHostToDevice, st));
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(R[2].ptr, ident.data(), rs[2], hipMemcpyHostToDevice, st));
    ECAL_HIP_TRY(ctx, hipStreamSynchronize(st));   // (pageable sources: consumed)
    if ((rc = ecal_rectify_batch_dev(ctx, (const double *) B[5].ptr, (const uint32_t *) B[6].ptr, (const uint32_t *) B[7].ptr,
                                     (const int32_t *) B[11].ptr, (const uint32_t *) B[13].ptr, (const uint32_t *) R[2].ptr,
                                     (const double *) R[0].ptr, F, (const double *) R[1].ptr, rprm, (double *) R[3].ptr, (uint32_t *) R[4].ptr,
                                     (uint32_t *) R[5].ptr, st)))
        return rc;
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(feat_xyr, R[3].ptr, rs[3], hipMemcpyDeviceToHost, st));
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(feat_valid, R[4].ptr, rs[4], hipMemcpyDeviceToHost, st));
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(frame_info, R[5].ptr, rs[5], hipMemcpyDeviceToHost, st));
    ECAL_HIP_TRY(ctx, hipStreamSynchronize(st));
    return ECAL_OK;
}

// ---- double-buffered ingest ------------------------------------------------------------------------------
// The event file does not have to be resident before detection starts: chunks of whole windows are uploaded on a
// copy stream (hipMemcpyAsync from pinned host memory) into one of two device buffers while the detection kernels
// of the previous chunk run on the context's stream.  Tiled windows (policy P1 of SURVEY 8d): window s =
// [t_start + s len, nextafter(t_start + (s + 1) len, -inf)], so every event is in exactly one window.
namespace {
__global__ void gather_features_kernel(const uint32_t *win_info, const uint32_t *seg_off, const double *cand_xyr,
                                       const int32_t *order, const uint32_t *found, uint32_t S, uint32_t M, double *feat) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * M) return;
    const uint32_t s = i / M;
    double x = NAN, y = NAN, r = NAN;
    if (ECAL_WIN_STATUS(win_info[4 * s + 3]) == 0 && found[s]) {
        const size_t c = (size_t) seg_off[2 * s] + (uint32_t) order[i];
        x = cand_xyr[3 * c];
        y = cand_xyr[3 * c + 1];
        r = cand_xyr[3 * c + 2];
    }
    feat[3 * (size_t) i] = x;
    feat[3 * (size_t) i + 1] = y;
    feat[3 * (size_t) i + 2] = r;
}
inline double rec_time(const uint8_t *events, uint64_t i) {
    double t;
    memcpy(&t, events + 25 * i, 8);
    return t;
}
}  // namespace

// ordered circles of every window on the device: d_feat[s][k] = candidate d_order[s][k] of window s (centre x, y,
// radius), NaN where the window has no complete grid — the keyframe features without a host round trip per window
extern "C" int ecal_gather_features_dev(ecal_ctx *ctx, const uint32_t *d_win_info, const uint32_t *d_seg_off, const double *d_cand_xyr,
                                        const int32_t *d_order, const uint32_t *d_found, uint32_t S, uint32_t M, double *d_feat,
                                        void *stream) {
    if (!ctx || (S && M && (!d_win_info || !d_seg_off || !d_cand_xyr || !d_order || !d_found || !d_feat))) return ECAL_ERR_INVALID;
    if (S == 0 || M == 0) return ECAL_OK;
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t tot = S * M;
    hipLaunchKernelGGL(gather_features_kernel, dim3((tot + 255) / 256), dim3(256), 0, (hipStream_t) stream, d_win_info, d_seg_off,
                       d_cand_xyr, d_order, d_found, S, M, d_feat);
    ECAL_HIP_TRY(ctx, hipGetLastError());
    return ECAL_OK;
}

// ---- one lock-step pass of the adaptive-window driver ------------------------------------------------------------
namespace {
__global__ void pack_pass_kernel(const uint32_t *win_info, const uint32_t *seg_off, const uint32_t *seg_cnt, const double *cand_xyr,
                                 const int32_t *order, const uint32_t *found, uint32_t S, uint32_t M, double *out) {
    const uint32_t W = 3 + 3 * M;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * (M + 1)) return;
    const uint32_t s = i / (M + 1), k = i % (M + 1);
    const bool ok = ECAL_WIN_STATUS(win_info[4 * s + 3]) == 0 && found[s];
    double *o = out + (size_t) s * W;
    if (k == M) {  // header: status, grid flag, EventFrame::eventsNum()
        o[0] = (double) win_info[4 * s + 3];
        o[1] = ok ? 1.0 : 0.0;
        o[2] = (double) (seg_cnt[2 * s] + seg_cnt[2 * s + 1]);
        return;
    }
    double x = NAN, y = NAN, r = NAN;
    if (ok) {
        const size_t c = (size_t) seg_off[2 * s] + (uint32_t) order[(size_t) s * M + k];
        x = cand_xyr[3 * c];
        y = cand_xyr[3 * c + 1];
        r = cand_xyr[3 * c + 2];
    }
    o[3 + 3 * k] = x;
    o[4 + 3 * k] = y;
    o[5 + 3 * k] = r;
}
}  // namespace

extern "C" int ecal_detect_pass(ecal_ctx *ctx, const uint8_t *d_events, uint64_t n_events, const double *t0, const double *t1,
                                uint32_t S, const ecal_detect_params *prm, uint32_t cap_points, double *packed) {
    if (!ctx || !prm || !packed || (S && (!t0 || !t1)) || (n_events && !d_events)) return ECAL_ERR_INVALID;
    const uint32_t M = prm->rows * prm->cols;
    if (S == 0) return ECAL_OK;
    if (M == 0 || M > 128) return ECAL_ERR_INVALID;
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    int rc;
    const size_t cap = (size_t) cap_points + 16, W = 3 + 3 * (size_t) M;
    ecal_devbuf *B = ctx->host_pipe;  // roles as in ecal_detect_batch; 0 holds t0 and t1 back to back
    const size_t sizes[17] = {2ul * S * sizeof(double), 16, S * 4ul, S * 4ul, (S + 1) * 4ul, cap * 16, 2ul * S * 4, 2ul * S * 4, cap * 4,
                              cap * 4, 2ul * S * 4, cap * 4, cap * 4, 4ul * S * 4, cap * 8, cap * 24, 16};
    for (int i = 0; i < 17; i++)
        if ((rc = ecal_ensure(ctx, B[i], sizes[i]))) return rc;
    if ((rc = ecal_ensure(ctx, ctx->host_grid_order, (size_t) S * M * sizeof(int32_t)))) return rc;
    if ((rc = ecal_ensure(ctx, ctx->host_grid_found, (size_t) S * sizeof(uint32_t)))) return rc;
    if ((rc = ecal_ensure(ctx, ctx->ingest_feat, (size_t) S * W * sizeof(double)))) return rc;
    const size_t pin_need = 2ul * S * sizeof(double) + (size_t) S * W * sizeof(double) + 16;
    if (ctx->pass_pinned_cap < pin_need) {
        if (ctx->pass_pinned) (void) hipHostFree(ctx->pass_pinned);
        ctx->pass_pinned = nullptr;
        ctx->pass_pinned_cap = 0;
        ECAL_HIP_TRY(ctx, hipHostMalloc((void **) &ctx->pass_pinned, pin_need + pin_need / 2, hipHostMallocDefault));
        ctx->pass_pinned_cap = pin_need + pin_need / 2;
    }
    double *h_t = ctx->pass_pinned, *h_out = ctx->pass_pinned + 2 * (size_t) S;
    memcpy(h_t, t0, S * sizeof(double));
    memcpy(h_t + S, t1, S * sizeof(double));
    double *d_t0 = (double *) B[0].ptr, *d_t1 = d_t0 + S;
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(d_t0, h_t, 2ul * S * sizeof(double), hipMemcpyHostToDevice, st));
    if ((rc = ecal_window_bounds_dev(ctx, d_events, n_events, d_t0, d_t1, S, (uint32_t *) B[2].ptr, (uint32_t *) B[3].ptr,
                                     (uint32_t *) B[4].ptr, st)))
        return rc;
    if ((rc = ecal_slice_events_dev(ctx, d_events, n_events, (uint32_t *) B[2].ptr, (uint32_t *) B[3].ptr, (uint32_t *) B[4].ptr, S, 0,
                                    cap_points, (double *) B[5].ptr, (uint32_t *) B[6].ptr, (uint32_t *) B[7].ptr, (int32_t *) B[8].ptr,
                                    (int *) B[16].ptr, st)))
        return rc;
    if ((rc = ecal_dbscan_batch_dev(ctx, (double *) B[5].ptr, (uint32_t *) B[6].ptr, (uint32_t *) B[7].ptr, 2 * S, cap_points, 0,
                                    prm->dbscan_eps, prm->dbscan_min_samples, (int32_t *) B[9].ptr, (uint32_t *) B[10].ptr, st)))
        return rc;
    if ((rc = ecal_extract_for_ctx(ctx, (double *) B[5].ptr, (uint32_t *) B[6].ptr, (uint32_t *) B[7].ptr, (int32_t *) B[9].ptr,
                                     (uint32_t *) B[10].ptr, S, cap_points, prm->dbscan_eps, prm->cluster_min_sample, prm->need_clusters,
                                     prm->circle_radius_threshold, prm->fit_circle, prm->knn_num, (uint32_t *) B[13].ptr,
                                     (uint32_t *) B[14].ptr, (double *) B[15].ptr, (int32_t *) B[11].ptr, (uint32_t *) B[12].ptr, st)))
        return rc;
    if ((rc = ecal_grid_order_dev(ctx, (uint32_t *) B[13].ptr, (uint32_t *) B[6].ptr, (double *) B[15].ptr, S, prm->rows, prm->cols,
                                  (int32_t *) ctx->host_grid_order.ptr, (uint32_t *) ctx->host_grid_found.ptr, st)))
        return rc;
    const uint32_t tot = S * (M + 1);
    hipLaunchKernelGGL(pack_pass_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, (const uint32_t *) B[13].ptr, (const uint32_t *) B[6].ptr,
                       (const uint32_t *) B[7].ptr, (const double *) B[15].ptr, (const int32_t *) ctx->host_grid_order.ptr,
                       (const uint32_t *) ctx->host_grid_found.ptr, S, M, (double *) ctx->ingest_feat.ptr);
    // the overflow flag lands in the pinned block too: a copy into pageable memory would take the runtime's synchronous path
    int *h_flag = reinterpret_cast<int *>(h_out + (size_t) S * W);
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(h_out, ctx->ingest_feat.ptr, (size_t) S * W * sizeof(double), hipMemcpyDeviceToHost, st));
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(h_flag, B[16].ptr, sizeof(int), hipMemcpyDeviceToHost, st));
    ECAL_HIP_TRY(ctx, hipStreamSynchronize(st));
    const int overflow = *h_flag;
    if (overflow) {
        ctx->last_error = "cap_points is smaller than the number of events covered by the windows";
        return ECAL_ERR_RANGE;
    }
    memcpy(packed, h_out, (size_t) S * W * sizeof(double));
    return ECAL_OK;
}

extern "C" int ecal_pin_host(ecal_ctx *ctx, void *ptr, size_t bytes) {
    if (!ctx || !ptr) return ECAL_ERR_INVALID;
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    ECAL_HIP_TRY(ctx, hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return ECAL_OK;
}
extern "C" int ecal_unpin_host(ecal_ctx *ctx, void *ptr) {
    if (!ctx || !ptr) return ECAL_ERR_INVALID;
    ECAL_HIP_TRY(ctx, hipHostUnregister(ptr));
    return ECAL_OK;
}

extern "C" int ecal_detect_stream_tiled(ecal_ctx *ctx, const uint8_t *events, uint64_t n_events, double t_start, double window_len,
                                        uint32_t windows_per_chunk, const ecal_detect_params *prm, uint32_t max_windows,
                                        uint32_t *win_info, uint32_t *grid_found, double *features, uint32_t *n_windows,
                                        ecal_ingest_stats *stats) {
    if (!ctx || !prm || !n_windows || (n_events && !events) || !(window_len > 0) || windows_per_chunk == 0) return ECAL_ERR_INVALID;
    *n_windows = 0;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (n_events == 0) return ECAL_OK;
    if (n_events > 0xFFFFFFFFull) return ECAL_ERR_RANGE;
    const double t_last = rec_time(events, n_events - 1);
    if (!(t_last >= t_start)) return ECAL_OK;
    const uint64_t S64 = (uint64_t) floor((t_last - t_start) / window_len) + 1;
    if (S64 > max_windows) {
        ctx->last_error = "ecal_detect_stream_tiled: more windows than max_windows";
        return ECAL_ERR_RANGE;
    }
    const uint32_t S = (uint32_t) S64, M = prm->rows * prm->cols;
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (!ctx->copy_stream) ECAL_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    hipStream_t cs = ctx->copy_stream;
    for (int k = 0; k < 2; k++) {
        if (!ctx->ev_uploaded[k]) ECAL_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_uploaded[k], hipEventDisableTiming));
        if (!ctx->ev_consumed[k]) ECAL_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_consumed[k], hipEventDisableTiming));
    }
    const auto wall0 = std::chrono::steady_clock::now();
    auto first_at_or_after = [&](double t) {  // lower_bound on the packed timestamps (host)
        uint64_t lo = 0, hi = n_events;
        while (lo < hi) {
            const uint64_t mid = (lo + hi) / 2;
            if (rec_time(events, mid) < t) lo = mid + 1;
            else hi = mid;
        }
        return lo;
    };
    const uint32_t n_chunks = (S + windows_per_chunk - 1) / windows_per_chunk;
    std::vector<uint64_t> c_lo(n_chunks), c_hi(n_chunks);
    uint64_t max_ev = 0;
    for (uint32_t c = 0; c < n_chunks; c++) {
        c_lo[c] = first_at_or_after(t_start + window_len * ((double) c * windows_per_chunk));
        const uint32_t w_end = std::min<uint64_t>(S, (uint64_t) (c + 1) * windows_per_chunk);
        c_hi[c] = w_end == S ? n_events : first_at_or_after(t_start + window_len * (double) w_end);
        max_ev = std::max(max_ev, c_hi[c] - c_lo[c]);
    }
    int rc;
    for (int k = 0; k < 2; k++)
        if ((rc = ecal_ensure(ctx, ctx->ingest_ev[k], max_ev * 25 + 32))) return rc;
    const uint32_t Wc = windows_per_chunk;
    const size_t cap = (size_t) max_ev + 16;
    ecal_devbuf *B = ctx->host_pipe;  // same roles as in ecal_detect_batch
    const size_t sizes[17] = {Wc * sizeof(double), Wc * sizeof(double), Wc * 4ul, Wc * 4ul, (Wc + 1) * 4ul, cap * 16, 2ul * Wc * 4,
                              2ul * Wc * 4, cap * 4, cap * 4, 2ul * Wc * 4, cap * 4, cap * 4, 4ul * Wc * 4, cap * 8, cap * 24, 16};
    for (int i = 0; i < 17; i++)
        if ((rc = ecal_ensure(ctx, B[i], sizes[i]))) return rc;
    if ((rc = ecal_ensure(ctx, ctx->host_grid_order, (size_t) Wc * (M ? M : 1) * sizeof(int32_t)))) return rc;
    if ((rc = ecal_ensure(ctx, ctx->host_grid_found, (size_t) Wc * sizeof(uint32_t)))) return rc;
    if ((rc = ecal_ensure(ctx, ctx->ingest_feat, (size_t) Wc * (M ? M : 1) * 24))) return rc;
    std::vector<double> t0(Wc), t1(Wc);
    auto upload = [&](uint32_t c) -> int {
        const int k = c & 1;
        if (c >= 2) ECAL_HIP_TRY(ctx, hipStreamWaitEvent(cs, ctx->ev_consumed[k], 0));  // buffer k is free again
        const uint64_t nb = (c_hi[c] - c_lo[c]) * 25;
        if (nb) ECAL_HIP_TRY(ctx, hipMemcpyAsync(ctx->ingest_ev[k].ptr, events + 25 * c_lo[c], nb, hipMemcpyHostToDevice, cs));
        ECAL_HIP_TRY(ctx, hipEventRecord(ctx->ev_uploaded[k], cs));
        return ECAL_OK;
    };
    if ((rc = upload(0))) return rc;
    for (uint32_t c = 0; c < n_chunks; c++) {
        if (c + 1 < n_chunks && (rc = upload(c + 1))) return rc;  // next chunk's copy overlaps this chunk's kernels
        const int k = c & 1;
        const uint32_t w0 = c * Wc, nw = std::min<uint64_t>(S, (uint64_t) (c + 1) * Wc) - w0;
        const uint64_t ne = c_hi[c] - c_lo[c];
        const uint8_t *d_ev = (const uint8_t *) ctx->ingest_ev[k].ptr;
        for (uint32_t w = 0; w < nw; w++) {
            t0[w] = t_start + window_len * (double) (w0 + w);
            t1[w] = nextafter(t_start + window_len * (double) (w0 + w + 1), -INFINITY);
        }
        // the host vectors are reused per chunk: wait until the previous chunk's copies of them were taken
        ECAL_HIP_TRY(ctx, hipMemcpyAsync(B[0].ptr, t0.data(), nw * sizeof(double), hipMemcpyHostToDevice, st));
        ECAL_HIP_TRY(ctx, hipMemcpyAsync(B[1].ptr, t1.data(), nw * sizeof(double), hipMemcpyHostToDevice, st));
        ECAL_HIP_TRY(ctx, hipStreamSynchronize(st));  // also bounds the host's run-ahead to one chunk
        ECAL_HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_uploaded[k], 0));
        if ((rc = ecal_window_bounds_dev(ctx, d_ev, ne, (double *) B[0].ptr, (double *) B[1].ptr, nw, (uint32_t *) B[2].ptr,
                                         (uint32_t *) B[3].ptr, (uint32_t *) B[4].ptr, st)))
            return rc;
        if ((rc = ecal_slice_events_dev(ctx, d_ev, ne, (uint32_t *) B[2].ptr, (uint32_t *) B[3].ptr, (uint32_t *) B[4].ptr, nw, 0,
                                        (uint32_t) max_ev, (double *) B[5].ptr, (uint32_t *) B[6].ptr, (uint32_t *) B[7].ptr,
                                        (int32_t *) B[8].ptr, (int *) B[16].ptr, st)))
            return rc;
        ECAL_HIP_TRY(ctx, hipEventRecord(ctx->ev_consumed[k], st));  // the packed records are not read after slicing
        if ((rc = ecal_dbscan_batch_dev(ctx, (double *) B[5].ptr, (uint32_t *) B[6].ptr, (uint32_t *) B[7].ptr, 2 * nw,
                                        (uint32_t) max_ev, 0, prm->dbscan_eps, prm->dbscan_min_samples, (int32_t *) B[9].ptr,
                                        (uint32_t *) B[10].ptr, st)))
            return rc;
        if ((rc = ecal_extract_for_ctx(ctx, (double *) B[5].ptr, (uint32_t *) B[6].ptr, (uint32_t *) B[7].ptr, (int32_t *) B[9].ptr,
                                         (uint32_t *) B[10].ptr, nw, (uint32_t) max_ev, prm->dbscan_eps, prm->cluster_min_sample, prm->need_clusters,
                                         prm->circle_radius_threshold, prm->fit_circle, prm->knn_num, (uint32_t *) B[13].ptr,
                                         (uint32_t *) B[14].ptr, (double *) B[15].ptr, (int32_t *) B[11].ptr, (uint32_t *) B[12].ptr,
                                         st)))
            return rc;
        if (win_info) ECAL_HIP_TRY(ctx, hipMemcpyAsync(win_info + 4 * (size_t) w0, B[13].ptr, 4ul * nw * 4, hipMemcpyDeviceToHost, st));
        if (M > 0) {
            if ((rc = ecal_grid_order_dev(ctx, (uint32_t *) B[13].ptr, (uint32_t *) B[6].ptr, (double *) B[15].ptr, nw, prm->rows,
                                          prm->cols, (int32_t *) ctx->host_grid_order.ptr, (uint32_t *) ctx->host_grid_found.ptr, st)))
                return rc;
            if (grid_found)
                ECAL_HIP_TRY(ctx, hipMemcpyAsync(grid_found + w0, ctx->host_grid_found.ptr, nw * 4ul, hipMemcpyDeviceToHost, st));
            if (features) {
                const uint32_t tot = nw * M;
                hipLaunchKernelGGL(gather_features_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, (const uint32_t *) B[13].ptr,
                                   (const uint32_t *) B[6].ptr, (const double *) B[15].ptr, (const int32_t *) ctx->host_grid_order.ptr,
                                   (const uint32_t *) ctx->host_grid_found.ptr, nw, M, (double *) ctx->ingest_feat.ptr);
                ECAL_HIP_TRY(ctx, hipMemcpyAsync(features + 3 * (size_t) w0 * M, ctx->ingest_feat.ptr, (size_t) tot * 24,
                                                 hipMemcpyDeviceToHost, st));
            }
        }
    }
    int overflow = 0;
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(&overflow, B[16].ptr, sizeof(int), hipMemcpyDeviceToHost, st));
    ECAL_HIP_TRY(ctx, hipStreamSynchronize(st));
    ECAL_HIP_TRY(ctx, hipStreamSynchronize(cs));
    *n_windows = S;
    if (stats) {
        stats->chunks = n_chunks;
        stats->max_chunk_events = max_ev;
        stats->bytes_uploaded = n_events ? (c_hi[n_chunks - 1] - c_lo[0]) * 25 : 0;
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count();
    }
    if (overflow) {
        ctx->last_error = "ecal_detect_stream_tiled: internal capacity overflow";
        return ECAL_ERR_RANGE;
    }
    return ECAL_OK;
}

// device-to-device copy on a stream (lets the Python all-reduce hook move the solver's buffer in and
// out of a torch tensor without any other HIP binding)
extern "C" int ecal_copy_dev(ecal_ctx *ctx, void *d_dst, const void *d_src, size_t bytes, void *stream, int sync) {
    if (!ctx || (bytes && (!d_dst || !d_src))) return ECAL_ERR_INVALID;
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->px_tree_labels = nullptr;   // (the destination may be the arrays the exported kd-trees belong to: ecal_ctx::px_tree)
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

// ---- the sequential keyframe gates of the init stage (host only) ------------------------------------------------------------
extern "C" int ecal_pose_gates(uint32_t n_frames, const double *Rsw, const double *twb, const double *time, const uint8_t *pnp_ok,
                               const uint8_t *rect_ok, double motion_time_step, uint32_t *accepted, uint32_t *n_accepted,
                               uint32_t *n_discarded_by_check_pose, uint32_t *n_discarded_by_rectify) {
    if (!n_accepted || (n_frames && (!Rsw || !twb || !time || !pnp_ok || !rect_ok || !accepted)) || !(motion_time_step > 0)) return ECAL_ERR_INVALID;
    const double lim_t = (2.5e-1 / motion_time_step) * 2, lim_r = (5e-4 * M_PI) * 2 / motion_time_step;
    uint32_t na = 0, nc = 0, nr = 0;
    long last = -1;
    for (uint32_t f = 0; f < n_frames; f++) {
        bool pose = pnp_ok[f] != 0;
        if (pose && last >= 0) {   // EventCalibIni::checkPose against the last accepted frame
            const double dt = time[f] - time[last];
            const double d0 = twb[3 * f] - twb[3 * last], d1 = twb[3 * f + 1] - twb[3 * last + 1], d2 = twb[3 * f + 2] - twb[3 * last + 2];
            const double v_t = std::sqrt(d0 * d0 + d1 * d1 + d2 * d2) / dt;
            double tr = 0.0;
            for (int i = 0; i < 9; i++) tr += Rsw[9 * (size_t) f + i] * Rsw[9 * (size_t) last + i];
            const double c = (tr - 1) * 0.5;
            const double v_r = std::fabs(std::acos(std::min(1.0, std::max(-1.0, c))) / dt);
            pose = v_t < lim_t && v_r < lim_r;
        }
        if (!pose) {
            nc++;
        } else if (!rect_ok[f]) {
            nr++;
        } else {
            accepted[na++] = f;
            last = (long) f;
        }
    }
    *n_accepted = na;
    if (n_discarded_by_check_pose) *n_discarded_by_check_pose = nc;
    if (n_discarded_by_rectify) *n_discarded_by_rectify = nr;
    return ECAL_OK;
}
