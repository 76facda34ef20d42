// Event -> (keyframe, circle) association that builds the solver's residual records.
//
// Replaces the loop of EventCalibSpline::optimize (event_camera_calib/src/EventCalibSpline.cpp:158-192):
// for every event with spline.front <= t <= spline.back, the keyframe nearest in time (1-D nanoflann
// tree, :140-154,166) accepted if dt^2 < (5 * MotionTimeStep)^2 (:168), then
// CirclesEventFrame::findCenter (include/.../CirclesEventFrame.hpp:50-65): nearest circle centre of that
// keyframe, accepted if |dist - radius| < 5 px; the event then becomes one residual (pixel, time,
// landmark of that circle).  Output order = event (time) order, as relationContainer_ (:181-183).
// Ties (equidistant keyframes / centres) go to the smaller index; nanoflann's choice is unpinned.
#include "ecal_ctx.hpp"

#pragma clang fp contract(off)

namespace ecal {

constexpr int AS_T = 256;
constexpr int AS_PER = 4;  // events per thread -> 1024 events per block

__device__ __forceinline__ double load_f64_u(const uint8_t *p) {
    double v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

// the range r with ranges[2 r] <= t <= ranges[2 r + 1] (ascending, disjoint), or -1
__device__ __forceinline__ int range_of(double t, const double *ranges, uint32_t R) {
    uint32_t a = 0, b = R;   // first range whose end is not below t
    while (a < b) {
        const uint32_t m = (a + b) >> 1;
        if (ranges[2 * m + 1] < t) a = m + 1; else b = m;
    }
    return (a < R && t >= ranges[2 * a]) ? (int) a : -1;
}

// returns the circle index or -1
__device__ __forceinline__ int associate_one(double t, double x, double y, const double *kf_time, const double *circles,
                                             uint32_t K, uint32_t n_circ, double t_min, double t_max, double max_dt2,
                                             double edge_tol) {
    if (!(t >= t_min && t <= t_max) || K == 0) return -1;
    // nearest keyframe in time: first index with kf_time >= t, compare with its predecessor
    uint32_t a = 0, b = K;
    while (a < b) {
        const uint32_t m = (a + b) >> 1;
        if (kf_time[m] < t) a = m + 1; else b = m;
    }
    uint32_t k = a;
    if (a == K) k = K - 1;
    else if (a > 0) {
        const double d0 = t - kf_time[a - 1], d1 = kf_time[a] - t;
        if (d0 * d0 <= d1 * d1) k = a - 1;
    }
    const double dt = t - kf_time[k];
    if (!(dt * dt < max_dt2)) return -1;
    const double *c = circles + 3 * (size_t) k * n_circ;
    double best = 1.79769313486231570e308;
    int bi = -1;
    for (uint32_t i = 0; i < n_circ; i++) {
        const double dx = x - c[3 * i], dy = y - c[3 * i + 1];
        const double d2 = dx * dx + dy * dy;
        if (d2 < best) {
            best = d2;
            bi = (int) i;
        }
    }
    if (bi < 0) return -1;
    const double dis = __dsqrt_rn(best);
    return (fabs(dis - c[3 * bi + 2]) < edge_tol) ? bi : -1;
}

// The same with the keyframes a block's events can meet staged in LDS (associate_kernel): kt / circ hold the keyframes
// k_first .. k_first + n_st - 1, the first index with kf_time >= t lies in [a_lo, a_hi] (the block's own events bound it).
// Same comparisons on the same values as associate_one.
__device__ __forceinline__ int associate_one_staged(double t, double x, double y, const double *kt, const double *circ, uint32_t K,
                                                    uint32_t n_circ, uint32_t k_first, uint32_t a_lo, uint32_t a_hi, double t_min,
                                                    double t_max, double max_dt2, double edge_tol) {
    if (!(t >= t_min && t <= t_max) || K == 0) return -1;
    uint32_t a = a_lo;
    while (a < a_hi && kt[a - k_first] < t) a++;
    uint32_t k = a;
    if (a == K) k = K - 1;
    else if (a > 0) {
        const double d0 = t - kt[a - 1 - k_first], d1 = kt[a - k_first] - t;
        if (d0 * d0 <= d1 * d1) k = a - 1;
    }
    const double dt = t - kt[k - k_first];
    if (!(dt * dt < max_dt2)) return -1;
    const double *c = circ + 3 * (size_t) (k - k_first) * n_circ;
    double best = 1.79769313486231570e308;
    int bi = -1;
    for (uint32_t i = 0; i < n_circ; i++) {
        const double dx = x - c[3 * i], dy = y - c[3 * i + 1];
        const double d2 = dx * dx + dy * dy;
        if (d2 < best) {
            best = d2;
            bi = (int) i;
        }
    }
    if (bi < 0) return -1;
    const double dis = __dsqrt_rn(best);
    return (fabs(dis - c[3 * bi + 2]) < edge_tol) ? bi : -1;
}

// first index in [0, K] whose kf_time is not below t, by one wave: 64 probes a round (three rounds for thousands of keyframes
// where a thread's bisection takes thirteen dependent loads)
__device__ __forceinline__ uint32_t wave_lower_bound(const double *kf_time, uint32_t K, double t) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t lo = 0, hi = K;   // every index below lo is < t, every index from hi on is >= t
    while (lo < hi) {
        const uint32_t span = hi - lo, step = (span + 63u) / 64u;
        const uint32_t idx = lo + lane * step;
        const bool less = idx < hi && kf_time[idx] < t;
        const uint32_t c = (uint32_t) __popcll(__ballot(less));   // ascending times: the probes below t come first
        if (c == 0) {
            hi = lo;
        } else {
            const uint32_t nhi = lo + c * step < hi ? lo + c * step : hi;
            lo = lo + (c - 1u) * step + 1u;
            hi = nhi;
        }
    }
    return lo;
}

constexpr uint32_t AS_KF_LDS = 6, AS_CIRC_LDS = 64, AS_RNG_LDS = 32;   // keyframes / circles per keyframe / time ranges a block stages

template <bool WRITE>
__global__ __launch_bounds__(AS_T) void associate_kernel(const uint8_t *__restrict__ rec, uint64_t n,
                                                         const double *__restrict__ kf_time,
                                                         const double *__restrict__ circles, uint32_t K,
                                                         uint32_t n_circ, double t_min, double t_max, double max_dt2,
                                                         double edge_tol, uint32_t *__restrict__ block_cnt,
                                                         const uint32_t *__restrict__ block_off,
                                                         double *__restrict__ obs, double *__restrict__ time,
                                                         uint32_t *__restrict__ lm, const double *__restrict__ ranges,
                                                         uint32_t n_ranges, uint32_t *__restrict__ seg) {
    __shared__ uint32_t wsum[AS_T / 64];
    __shared__ double s_kt[AS_KF_LDS], s_circ[AS_KF_LDS * 3 * AS_CIRC_LDS], s_rng[2 * AS_RNG_LDS], s_tmm[2 * (AS_T / 64)];
    __shared__ uint32_t s_plan[4];   // staged?, k_first, a_lo, a_hi
    const uint64_t base = (uint64_t) blockIdx.x * (AS_T * AS_PER) + (uint64_t) threadIdx.x * AS_PER;
    int hit[AS_PER], rng[AS_PER];
    double tt[AS_PER], xx[AS_PER], yy[AS_PER];
    uint32_t mine = 0;
    double tmn = 1.79769313486231570e308, tmx = -1.79769313486231570e308;
#pragma unroll
    for (int e = 0; e < AS_PER; e++) {
        hit[e] = -1;
        rng[e] = 0;
        tt[e] = xx[e] = yy[e] = 0.0;
        const uint64_t i = base + e;
        if (i < n) {
            const uint8_t *r = rec + i * 25;
            tt[e] = load_f64_u(r);
            xx[e] = load_f64_u(r + 8);
            yy[e] = load_f64_u(r + 16);
            tmn = tt[e] < tmn ? tt[e] : tmn;
            tmx = tt[e] > tmx ? tt[e] : tmx;
        }
    }
    // The keyframes this block's 1024 events can meet: the events lie close together in time (the stream is in time order: a
    // millisecond a block), so the nearest keyframe of every one of them is among the few around the block's own time span —
    // found once per block (a wave's 64-ary search) and staged in LDS with their circles, instead of a thirteen-step bisection
    // through global memory and 108 global loads of circle data PER EVENT.  Any order of events is taken (the span is the
    // block's minimum and maximum); a block whose span meets more than AS_KF_LDS keyframes goes the plain way.
    {
        const int lane_ = threadIdx.x & 63, wave_ = threadIdx.x >> 6;
        for (int o = 32; o > 0; o >>= 1) {
            const double a = __shfl_xor(tmn, o, 64), b = __shfl_xor(tmx, o, 64);
            tmn = a < tmn ? a : tmn;
            tmx = b > tmx ? b : tmx;
        }
        if (lane_ == 0) {
            s_tmm[2 * wave_] = tmn;
            s_tmm[2 * wave_ + 1] = tmx;
        }
        if (ranges && n_ranges <= AS_RNG_LDS && threadIdx.x < 2 * n_ranges) s_rng[threadIdx.x] = ranges[threadIdx.x];
        __syncthreads();
        if (wave_ == 0) {
            double bmn = s_tmm[0], bmx = s_tmm[1];
            for (int w = 1; w < AS_T / 64; w++) {
                bmn = s_tmm[2 * w] < bmn ? s_tmm[2 * w] : bmn;
                bmx = s_tmm[2 * w + 1] > bmx ? s_tmm[2 * w + 1] : bmx;
            }
            uint32_t staged = 0, k_first = 0, a_lo = 0, a_hi = 0;
            if (K > 0 && n_circ <= AS_CIRC_LDS && bmn <= bmx) {
                a_lo = wave_lower_bound(kf_time, K, bmn);
                a_hi = wave_lower_bound(kf_time, K, bmx);
                k_first = a_lo > 0 ? a_lo - 1 : 0;
                const uint32_t k_last = a_hi < K ? a_hi : K - 1;
                if (k_last - k_first + 1 <= AS_KF_LDS) {
                    staged = 1;
                    const uint32_t n_st = k_last - k_first + 1;
                    if ((uint32_t) lane_ < n_st) s_kt[lane_] = kf_time[k_first + lane_];
                    for (uint32_t i = lane_; i < n_st * 3 * n_circ; i += 64) s_circ[i] = circles[3 * (size_t) k_first * n_circ + i];
                }
            }
            if (lane_ == 0) {
                s_plan[0] = staged;
                s_plan[1] = k_first;
                s_plan[2] = a_lo;
                s_plan[3] = a_hi;
            }
        }
        __syncthreads();
    }
    const bool staged = s_plan[0] != 0;
    const uint32_t k_first = s_plan[1], a_lo = s_plan[2], a_hi = s_plan[3];
    const double *const rng_tab = ranges && n_ranges <= AS_RNG_LDS ? s_rng : ranges;
#pragma unroll
    for (int e = 0; e < AS_PER; e++) {
        const uint64_t i = base + e;
        if (i < n) {
            if (ranges) {   // all spline segments in one pass: the event's range, if any, names its segment
                rng[e] = range_of(tt[e], rng_tab, n_ranges);
                if (rng[e] >= 0)
                    hit[e] = staged ? associate_one_staged(tt[e], xx[e], yy[e], s_kt, s_circ, K, n_circ, k_first, a_lo, a_hi, tt[e], tt[e], max_dt2, edge_tol)
                                    : associate_one(tt[e], xx[e], yy[e], kf_time, circles, K, n_circ, tt[e], tt[e], max_dt2, edge_tol);
            } else {
                hit[e] = staged ? associate_one_staged(tt[e], xx[e], yy[e], s_kt, s_circ, K, n_circ, k_first, a_lo, a_hi, t_min, t_max, max_dt2, edge_tol)
                                : associate_one(tt[e], xx[e], yy[e], kf_time, circles, K, n_circ, t_min, t_max, max_dt2, edge_tol);
            }
            mine += hit[e] >= 0 ? 1u : 0u;
        }
    }
    // exclusive scan of `mine` over the block (thread order = event order)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t pre = 0, tot = 0;
    for (int w = 0; w < AS_T / 64; w++) {
        if (w < wave) pre += wsum[w];
        tot += wsum[w];
    }
    if (!WRITE) {
        if (threadIdx.x == 0) block_cnt[blockIdx.x] = tot;
        return;
    }
    uint64_t at = (uint64_t) block_off[blockIdx.x] + pre + inc - mine;
#pragma unroll
    for (int e = 0; e < AS_PER; e++) {
        if (hit[e] >= 0) {
            obs[2 * at] = xx[e];
            obs[2 * at + 1] = yy[e];
            time[at] = tt[e];
            lm[at] = (uint32_t) hit[e];
            if (seg) seg[at] = (uint32_t) rng[e];
            at++;
        }
    }
}

// exclusive scan of the per-block counts; off[n_blocks] = total.  One workgroup.
__global__ __launch_bounds__(1024) void scan_blocks_kernel(const uint32_t *__restrict__ cnt, uint32_t nb,
                                                           uint32_t *__restrict__ off) {
    __shared__ uint32_t red[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < nb; b0 += 1024) {
        const uint32_t b = b0 + threadIdx.x;
        const uint32_t v = b < nb ? cnt[b] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        if (lane == 63) red[wave] = inc;
        __syncthreads();
        uint32_t pre = 0, tot = 0;
        for (int w = 0; w < 16; w++) {
            if (w < wave) pre += red[w];
            tot += red[w];
        }
        if (b < nb) off[b] = carry + pre + inc - v;
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) off[nb] = carry;
}

}  // namespace ecal

using namespace ecal;

static int associate_common(ecal_ctx *ctx, const uint8_t *d_events, uint64_t n_events, const double *d_kf_time,
                            const double *d_kf_circles, uint32_t n_keyframes, uint32_t n_circles, double t_min, double t_max,
                            const double *d_ranges, uint32_t n_ranges, double max_dt, double edge_tol, double *d_obs, double *d_time,
                            uint32_t *d_lm_id, uint32_t *d_seg_id, uint32_t *d_count, void *stream) {
    if (!ctx || !d_count) return ECAL_ERR_INVALID;
    if (n_events > 0xFFFFFFFFull) return ECAL_ERR_RANGE;
    if (n_events && (!d_events || !d_obs || !d_time || !d_lm_id)) return ECAL_ERR_INVALID;
    if (n_keyframes && (!d_kf_time || !d_kf_circles)) return ECAL_ERR_INVALID;
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t) stream;
    const uint32_t nb = (uint32_t) ((n_events + AS_T * AS_PER - 1) / (AS_T * AS_PER));
    if (nb == 0) {
        ECAL_HIP_TRY(ctx, hipMemsetAsync(d_count, 0, sizeof(uint32_t), st));
        return ECAL_OK;
    }
    int rc;
    if ((rc = ecal_ensure(ctx, ctx->as_cnt, (size_t) nb * sizeof(uint32_t)))) return rc;
    if ((rc = ecal_ensure(ctx, ctx->as_off, ((size_t) nb + 1) * sizeof(uint32_t)))) return rc;
    uint32_t *cnt = (uint32_t *) ctx->as_cnt.ptr, *off = (uint32_t *) ctx->as_off.ptr;
    const double md2 = max_dt * max_dt;
    hipLaunchKernelGGL((associate_kernel<false>), dim3(nb), dim3(AS_T), 0, st, d_events, n_events, d_kf_time,
                       d_kf_circles, n_keyframes, n_circles, t_min, t_max, md2, edge_tol, cnt, off, d_obs, d_time,
                       d_lm_id, d_ranges, n_ranges, d_seg_id);
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1024), 0, st, cnt, nb, off);
    hipLaunchKernelGGL((associate_kernel<true>), dim3(nb), dim3(AS_T), 0, st, d_events, n_events, d_kf_time,
                       d_kf_circles, n_keyframes, n_circles, t_min, t_max, md2, edge_tol, cnt, off, d_obs, d_time,
                       d_lm_id, d_ranges, n_ranges, d_seg_id);
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(d_count, off + nb, sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
    ECAL_HIP_TRY(ctx, hipGetLastError());
    return ECAL_OK;
}

extern "C" int ecal_associate_dev(ecal_ctx *ctx, const uint8_t *d_events, uint64_t n_events, const double *d_kf_time,
                                  const double *d_kf_circles, uint32_t n_keyframes, uint32_t n_circles, double t_min,
                                  double t_max, double max_dt, double edge_tol, double *d_obs, double *d_time,
                                  uint32_t *d_lm_id, uint32_t *d_count, void *stream) {
    return associate_common(ctx, d_events, n_events, d_kf_time, d_kf_circles, n_keyframes, n_circles, t_min, t_max, nullptr, 0, max_dt,
                            edge_tol, d_obs, d_time, d_lm_id, nullptr, d_count, stream);
}

extern "C" int ecal_associate_ranges_dev(ecal_ctx *ctx, const uint8_t *d_events, uint64_t n_events, const double *d_kf_time,
                                         const double *d_kf_circles, uint32_t n_keyframes, uint32_t n_circles, const double *d_ranges,
                                         uint32_t n_ranges, double max_dt, double edge_tol, double *d_obs, double *d_time,
                                         uint32_t *d_lm_id, uint32_t *d_seg_id, uint32_t *d_count, void *stream) {
    if (ctx && n_events && (!d_ranges || !d_seg_id) && n_ranges) {
        ctx->last_error = "null pointer";
        return ECAL_ERR_INVALID;
    }
    if (n_ranges == 0) {   // no spline segment: no residual
        if (!ctx || !d_count) return ECAL_ERR_INVALID;
        ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
        ECAL_HIP_TRY(ctx, hipMemsetAsync(d_count, 0, sizeof(uint32_t), (hipStream_t) stream));
        return ECAL_OK;
    }
    return associate_common(ctx, d_events, n_events, d_kf_time, d_kf_circles, n_keyframes, n_circles, 0.0, 0.0, d_ranges, n_ranges, max_dt,
                            edge_tol, d_obs, d_time, d_lm_id, d_seg_id, d_count, stream);
}

// host-buffer form (what host/event_calib_spline.hpp calls): the event stream stays in HBM (ecal_stream), the
// keyframe tables go up, the accepted records come back.  obs / time / lm_id need room for `capacity` records;
// *count is the number found (ECAL_ERR_RANGE and nothing copied if it exceeds capacity).
extern "C" int ecal_associate(ecal_ctx *ctx, const ecal_stream *es, const double *kf_time, const double *kf_circles,
                              uint32_t n_keyframes, uint32_t n_circles, double t_min, double t_max, double max_dt, double edge_tol,
                              uint64_t capacity, double *obs, double *time, uint32_t *lm_id, uint64_t *count) {
    if (!ctx || !es || !count) return ECAL_ERR_INVALID;
    const uint64_t n = ecal_stream_size(es);
    *count = 0;
    if (n == 0 || n_keyframes == 0) return ECAL_OK;
    if (!kf_time || !kf_circles) return ECAL_ERR_INVALID;
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t kb = (size_t) n_keyframes * 8, cb = (size_t) n_keyframes * n_circles * 24;
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t o_kt = 0, o_kc = up(kb), o_obs = o_kc + up(cb), o_tm = o_obs + up(n * 16), o_lm = o_tm + up(n * 8),
                 o_cnt = o_lm + up(n * 4), total = o_cnt + 256;
    int rc = ecal_ensure(ctx, ctx->as_host, total);
    if (rc) return rc;
    char *base = (char *) ctx->as_host.ptr;
    hipStream_t st = ctx->stream;
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(base + o_kt, kf_time, kb, hipMemcpyHostToDevice, st));
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(base + o_kc, kf_circles, cb, hipMemcpyHostToDevice, st));
    rc = ecal_associate_dev(ctx, ecal_stream_data(es), n, (const double *) (base + o_kt), (const double *) (base + o_kc), n_keyframes,
                            n_circles, t_min, t_max, max_dt, edge_tol, (double *) (base + o_obs), (double *) (base + o_tm),
                            (uint32_t *) (base + o_lm), (uint32_t *) (base + o_cnt), st);
    if (rc) return rc;
    uint32_t cnt = 0;
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(&cnt, base + o_cnt, 4, hipMemcpyDeviceToHost, st));
    ECAL_HIP_TRY(ctx, hipStreamSynchronize(st));
    *count = cnt;
    if (cnt > capacity) return ECAL_ERR_RANGE;
    if (cnt) {
        if (!obs || !time || !lm_id) return ECAL_ERR_INVALID;
        ECAL_HIP_TRY(ctx, hipMemcpyAsync(obs, base + o_obs, (size_t) cnt * 16, hipMemcpyDeviceToHost, st));
        ECAL_HIP_TRY(ctx, hipMemcpyAsync(time, base + o_tm, (size_t) cnt * 8, hipMemcpyDeviceToHost, st));
        ECAL_HIP_TRY(ctx, hipMemcpyAsync(lm_id, base + o_lm, (size_t) cnt * 4, hipMemcpyDeviceToHost, st));
        ECAL_HIP_TRY(ctx, hipStreamSynchronize(st));
    }
    return ECAL_OK;
}


// Association of every spline segment + the solver built on the result, in one host call (what host/event_calib_spline.hpp
// calls): the event stream is resident (ecal_stream), the keyframe tables and the segments' time ranges go up, the residual
// arrays never leave HBM (ecal_associate_ranges_dev -> ecal_solver_create_dev).  layout = the problem without its residual
// arrays (obs / time / lm_id / seg_id / n_res are ignored).
extern "C" int ecal_solver_create_from_stream(ecal_ctx *ctx, const ecal_stream *es, const double *kf_time, const double *kf_circles,
                                              uint32_t n_keyframes, uint32_t n_circles, const double *ranges, uint32_t n_ranges,
                                              double max_dt, double edge_tol, const ecal_spline_problem *layout, ecal_solver **out) {
    if (!ctx || !es || !layout || !out || (n_keyframes && (!kf_time || !kf_circles)) || (n_ranges && !ranges)) return ECAL_ERR_INVALID;
    *out = nullptr;
    if (n_ranges != layout->n_segments) {
        ctx->last_error = "ecal_solver_create_from_stream: one time range per spline segment";
        return ECAL_ERR_INVALID;
    }
    // the device side finds an event's range by bisection (range_of): ascending, disjoint, well-formed ranges or nothing
    for (uint32_t r = 0; r < n_ranges; r++) {
        const bool ok = ranges[2 * r] <= ranges[2 * r + 1] && (r == 0 || ranges[2 * r - 1] < ranges[2 * r]);
        if (!ok) {   // (also catches NaN bounds)
            ctx->last_error = "ecal_solver_create_from_stream: the time ranges must be ascending and disjoint (range " + std::to_string(r) + ")";
            return ECAL_ERR_INVALID;
        }
    }
    const uint64_t n = ecal_stream_size(es);
    if (n > 0xFFFFFFFFull) return ECAL_ERR_RANGE;
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t kb = (size_t) n_keyframes * 8, cb = (size_t) n_keyframes * n_circles * 24, rb = (size_t) n_ranges * 16;
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t o_kt = 0, o_kc = up(kb), o_rg = o_kc + up(cb), o_obs = o_rg + up(rb), o_tm = o_obs + up(n * 16), o_lm = o_tm + up(n * 8),
                 o_sg = o_lm + up(n * 4), o_cnt = o_sg + up(n * 4), total = o_cnt + 256;
    int rc = ecal_ensure(ctx, ctx->as_host, total);
    if (rc) return rc;
    char *base = (char *) ctx->as_host.ptr;
    hipStream_t st = ctx->stream;
    if (kb) ECAL_HIP_TRY(ctx, hipMemcpyAsync(base + o_kt, kf_time, kb, hipMemcpyHostToDevice, st));
    if (cb) ECAL_HIP_TRY(ctx, hipMemcpyAsync(base + o_kc, kf_circles, cb, hipMemcpyHostToDevice, st));
    if (rb) ECAL_HIP_TRY(ctx, hipMemcpyAsync(base + o_rg, ranges, rb, hipMemcpyHostToDevice, st));
    ECAL_HIP_TRY(ctx, hipStreamSynchronize(st));   // (pageable sources: consumed)
    rc = ecal_associate_ranges_dev(ctx, ecal_stream_data(es), n, (const double *) (base + o_kt), (const double *) (base + o_kc), n_keyframes,
                                   n_circles, (const double *) (base + o_rg), n_ranges, max_dt, edge_tol, (double *) (base + o_obs),
                                   (double *) (base + o_tm), (uint32_t *) (base + o_lm), (uint32_t *) (base + o_sg),
                                   (uint32_t *) (base + o_cnt), st);
    if (rc) return rc;
    ecal_spline_problem p = *layout;
    p.obs = (const double *) (base + o_obs);
    p.time = (const double *) (base + o_tm);
    p.lm_id = (const uint32_t *) (base + o_lm);
    p.seg_id = (const uint32_t *) (base + o_sg);
    p.n_res = n;
    return ecal_solver_create_dev(ctx, &p, (const uint32_t *) (base + o_cnt), st, out);
}
