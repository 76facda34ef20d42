// Context lifecycle and shared helpers of libecal.so.
#include <dlfcn.h>
#include "ecal_ctx.hpp"

constexpr uint32_t ZERO_RING_WORDS = 16384, ZERO_RING_GRAIN = 16;

void ecal_read_switches(ecal_switches &sw) {
    auto on = [](const char *name) { return getenv(name) != nullptr; };
    auto num = [](const char *name) -> long long {
        const char *e = getenv(name);
        return e ? atoll(e) : 0;
    };
    sw = ecal_switches();
    sw.slice_no_pixel = on("ECAL_SLICE_NO_PIXEL");
    sw.slice_sort_kernel = on("ECAL_SLICE_SORT_KERNEL");
    sw.slice_no_second_pass = on("ECAL_SLICE_NO_SECOND_PASS");
    sw.bounds_two_kernels = on("ECAL_BOUNDS_TWO_KERNELS");
    sw.dbscan_no_pixel = on("ECAL_DBSCAN_NO_PIXEL");
    sw.dbscan_generic_disc = on("ECAL_DBSCAN_GENERIC_DISC");
    sw.extract_no_inline_ties = on("ECAL_EXTRACT_NO_INLINE_TIES");
    sw.no_zero_ring = on("ECAL_NO_ZERO_RING");
    sw.adaptive_trace = on("ECAL_ADAPTIVE_TRACE");
    sw.adaptive_rounds = on("ECAL_ADAPTIVE_ROUNDS");
    sw.adaptive_deal_uniform = on("ECAL_ADAPTIVE_DEAL_UNIFORM");
    sw.grid_debug = on("ECAL_GRID_DEBUG");
    sw.grid_serial_walk = on("ECAL_GRID_SERIAL_WALK");
    sw.grid_one_wave = on("ECAL_GRID_ONE_WAVE");
    sw.solver_device_linear_solve = on("ECAL_SOLVER_DEVICE_LINEAR_SOLVE");
    sw.solver_trace = on("ECAL_SOLVER_TRACE");
    sw.solver_no_stream = on("ECAL_SOLVER_NO_STREAM");
    sw.adaptive_dir_kernel = on("ECAL_ADAPTIVE_DIR_KERNEL");
    sw.adaptive_verify_in_alloc = on("ECAL_ADAPTIVE_VERIFY_IN_ALLOC");
    sw.adaptive_depth = (int) num("ECAL_ADAPTIVE_DEPTH");
    sw.adaptive_depth_max = (int) num("ECAL_ADAPTIVE_DEPTH_MAX");
    sw.adaptive_live_floor = (int) num("ECAL_ADAPTIVE_LIVE_FLOOR");
    sw.adaptive_grid_pieces = (int) num("ECAL_ADAPTIVE_GRID_PIECES");
    if (getenv("ECAL_ADAPTIVE_SIDE")) sw.adaptive_side = (int) num("ECAL_ADAPTIVE_SIDE");
    if (getenv("ECAL_ADAPTIVE_TREE")) sw.adaptive_tree = (int) num("ECAL_ADAPTIVE_TREE");
    sw.arrow_k = (int) num("ECAL_ARROW_K");
    sw.bo_big_arena = (unsigned long long) num("ECAL_BO_BIG_ARENA");
    if (const char *e = getenv("ECAL_GRID_TOL_PX")) sw.grid_tol_px = atof(e);
}

// tests: the switches again, after the environment changed under a live context
extern "C" int ecal_debug_reload_env(ecal_ctx *ctx) {
    if (!ctx) return ECAL_ERR_INVALID;
    ecal_read_switches(ctx->sw);
    return ECAL_OK;
}

uint32_t *ecal_zero_words(ecal_ctx *ctx, hipStream_t st, uint32_t n) {
    if (n > ZERO_RING_GRAIN || ctx->sw.no_zero_ring) return nullptr;
    ecal_ctx::zero_ring *r = nullptr;
    for (auto &c : ctx->zero_rings)
        if (c.used && c.stream == st) r = &c;
    if (!r) {
        for (auto &c : ctx->zero_rings)
            if (!c.used && !r) r = &c;
        if (!r) return nullptr;
        if (hipMalloc((void **) &r->ptr, ZERO_RING_WORDS * sizeof(uint32_t)) != hipSuccess) {
            r->ptr = nullptr;
            return nullptr;
        }
        r->used = true;
        r->stream = st;
        r->pos = ZERO_RING_WORDS;   // wiped below
    }
    // The ring is wiped half by half, on entry: the words handed out last (the other half: 512 calls) stay as their owners
    // left them — an entry point may still be enqueueing kernels that read a counter it took a few calls ago —, and every
    // earlier user of the half entered is ahead of this memset on the same stream.
    constexpr uint32_t HALF = ZERO_RING_WORDS / 2;
    if (r->pos >= ZERO_RING_WORDS) r->pos = 0;
    if (r->pos % HALF == 0) {
        if (hipMemsetAsync(r->ptr + r->pos, 0, HALF * sizeof(uint32_t), st) != hipSuccess) return nullptr;
    }
    uint32_t *p = r->ptr + r->pos;
    r->pos += ZERO_RING_GRAIN;
    return p;
}

int ecal_ensure(ecal_ctx *ctx, ecal_devbuf &b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (b.cap >= bytes) return ECAL_OK;
    if (b.ptr) {
        hipError_t e = hipFree(b.ptr);
        b.ptr = nullptr;
        b.cap = 0;
        if (e != hipSuccess) {
            ctx->last_error = std::string("hipFree: ") + hipGetErrorString(e);
            return ECAL_ERR_HIP;
        }
    }
    // grow geometrically so that a stream of slowly growing batches does not re-allocate each time
    size_t want = bytes + bytes / 4;
    want = (want + 255) & ~(size_t) 255;
    hipError_t e = hipMalloc(&b.ptr, want);
    if (e != hipSuccess) {
        want = (bytes + 255) & ~(size_t) 255;
        e = hipMalloc(&b.ptr, want);
    }
    if (e != hipSuccess) {
        b.ptr = nullptr;
        ctx->last_error = std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e);
        return ECAL_ERR_NOMEM;
    }
    b.cap = want;
    return ECAL_OK;
}

extern "C" int ecal_abi_version(void) { return ECAL_ABI_VERSION; }

extern "C" const char *ecal_strerror(int status) {
    switch (status) {
        case ECAL_OK: return "ok";
        case ECAL_ERR_INVALID: return "invalid argument";
        case ECAL_ERR_NO_DEVICE: return "no usable HIP device";
        case ECAL_ERR_HIP: return "HIP runtime error";
        case ECAL_ERR_NOMEM: return "out of memory";
        case ECAL_ERR_UNSORTED: return "event timestamps not sorted";
        case ECAL_ERR_RANGE: return "size out of range";
        case ECAL_ERR_COMM: return "RCCL error";
        default: return "unknown status";
    }
}

extern "C" const char *ecal_last_error(const ecal_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

extern "C" int ecal_init(int device, ecal_ctx **out) {
    if (!out) return ECAL_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return ECAL_ERR_NO_DEVICE;
    if (device < 0 || device >= count) return ECAL_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return ECAL_ERR_NO_DEVICE;
    ecal_ctx *ctx = new (std::nothrow) ecal_ctx;
    if (!ctx) return ECAL_ERR_NOMEM;
    ctx->device = device;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return ECAL_ERR_HIP;
    }
    ecal_read_switches(ctx->sw);
    if (const char *e = getenv("ECAL_MEDIAN_TIES")) ctx->median_ties = atoi(e) ? ECAL_TIES_SMALLER_PID : ECAL_TIES_REFERENCE;   // debug switch
    if (getenv("ECAL_ROCTX")) (void) ecal_set_profile_ranges(ctx, 1);   // (no marker library: no ranges, not an error)
    if (const char *e = getenv("ECAL_TAIL_MODE")) ctx->tail_mode = atoi(e);   // debug switch (0 auto, 1 every tier, 2 lean)
    if (hipHostMalloc((void **) &ctx->tail_seen, ECAL_TAIL_SLOTS * sizeof(uint32_t), hipHostMallocMapped) == hipSuccess &&
        hipHostGetDevicePointer((void **) &ctx->tail_seen_dev, ctx->tail_seen, 0) == hipSuccess) {
        for (int k = 0; k < ECAL_TAIL_SLOTS; k++) ctx->tail_seen[k] = 0xFFFFFFFFu;
    } else {   // no mapped host memory: every tier every time
        if (ctx->tail_seen) (void) hipHostFree(ctx->tail_seen);
        ctx->tail_seen = ctx->tail_seen_dev = nullptr;
        (void) hipGetLastError();
    }
    *out = ctx;
    return ECAL_OK;
}

static void release(ecal_devbuf &b) {
    if (b.ptr) (void) hipFree(b.ptr);
    b.ptr = nullptr;
    b.cap = 0;
}

extern "C" void ecal_destroy(ecal_ctx *ctx) {
    if (!ctx) return;
    (void) hipSetDevice(ctx->device);
    if (ctx->stream) {
        (void) hipStreamSynchronize(ctx->stream);
        (void) hipStreamDestroy(ctx->stream);
    }
    (void) ecal_comm_destroy(ctx);
    if (ctx->calib_pinned) (void) hipHostFree(ctx->calib_pinned);
    if (ctx->tail_seen) (void) hipHostFree(ctx->tail_seen);
    if (ctx->copy_stream) (void) hipStreamDestroy(ctx->copy_stream);
    for (auto &c : ctx->zero_rings)
        if (c.ptr) (void) hipFree(c.ptr);
    for (auto &e : ctx->adaptive_ev)
        if (e) (void) hipEventDestroy(e);
    if (ctx->pass_pinned) (void) hipHostFree(ctx->pass_pinned);
    for (int k = 0; k < 2; k++) {
        if (ctx->ev_uploaded[k]) (void) hipEventDestroy(ctx->ev_uploaded[k]);
        if (ctx->ev_consumed[k]) (void) hipEventDestroy(ctx->ev_consumed[k]);
    }
    for (ecal_devbuf *b : ctx->all_bufs()) release(*b);
    ctx->roctx_push = nullptr;
    ctx->roctx_pop = nullptr;
    if (ctx->roctx_lib) (void) dlclose(ctx->roctx_lib);
    delete ctx;
}

extern "C" int ecal_set_profile_ranges(ecal_ctx *ctx, int on) {
    if (!ctx) return ECAL_ERR_INVALID;
    if (!on) {
        ctx->roctx_push = nullptr;
        ctx->roctx_pop = nullptr;
        return ECAL_OK;
    }
    if (!ctx->roctx_lib) {
        for (const char *name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            ctx->roctx_lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (ctx->roctx_lib) break;
        }
    }
    if (!ctx->roctx_lib) {
        ctx->last_error = "ecal_set_profile_ranges: no roctx library (librocprofiler-sdk-roctx.so / libroctx64.so) to be found";
        return ECAL_ERR_INVALID;
    }
    ctx->roctx_push = (int (*)(const char *)) dlsym(ctx->roctx_lib, "roctxRangePushA");
    ctx->roctx_pop = (int (*)()) dlsym(ctx->roctx_lib, "roctxRangePop");
    if (!ctx->roctx_push || !ctx->roctx_pop) {
        ctx->roctx_push = nullptr;
        ctx->roctx_pop = nullptr;
        ctx->last_error = "ecal_set_profile_ranges: roctxRangePushA / roctxRangePop not exported";
        return ECAL_ERR_INVALID;
    }
    return ECAL_OK;
}

// tests / tools: what the stages' last kernels reported (ecal_ctx::tail_seen, 16 words; 0xFFFFFFFF = nothing yet)
extern "C" int ecal_debug_tail_seen(ecal_ctx *ctx, uint32_t *out16) {
    if (!ctx || !out16 || !ctx->tail_seen) return ECAL_ERR_INVALID;
    for (int k = 0; k < ECAL_TAIL_SLOTS; k++) out16[k] = __atomic_load_n(ctx->tail_seen + k, __ATOMIC_RELAXED);
    return ECAL_OK;
}

extern "C" int ecal_get_tail_mode(const ecal_ctx *ctx) { return ctx ? ctx->tail_mode : ECAL_ERR_INVALID; }
extern "C" int ecal_set_tail_mode(ecal_ctx *ctx, int mode) {
    if (!ctx || mode < ECAL_TAIL_AUTO || mode > ECAL_TAIL_LEAN) return ECAL_ERR_INVALID;
    ctx->tail_mode = mode;
    if (ctx->tail_seen)   // what earlier calls saw says nothing about the calls to come in the new mode
        for (int k = 0; k < ECAL_TAIL_SLOTS; k++) ctx->tail_seen[k] = 0xFFFFFFFFu;
    return ECAL_OK;
}

extern "C" int ecal_sync(ecal_ctx *ctx) {
    if (!ctx) return ECAL_ERR_INVALID;
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    ECAL_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return ECAL_OK;
}
