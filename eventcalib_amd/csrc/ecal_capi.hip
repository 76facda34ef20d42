// Context lifecycle and shared helpers of libecal.so.
#include <dlfcn.h>
#include <cstring>
#include "ecal_ctx.hpp"

constexpr uint32_t ZERO_RING_WORDS = 16384, ZERO_RING_GRAIN = 16;

// The environment of the library, read once per context (ecal_init; ecal_debug_reload_env for tests):
//   ECAL_FORCE          test knobs, comma-separated: which tier / routine produces a result, never what the result is —
//                       slice_general, dbscan_general, dbscan_generic_disc, extract_no_inline_ties, bounds_two_kernels, grid_one_wave,
//                       grid_serial_walk, solver_no_stream, latency_two_pass, latency_forms (the parity tests run the forms against each other)
//   ECAL_TRACE          stderr traces, comma-separated: adaptive, grid, solver, load
//   ECAL_ADAPTIVE_SHAPE the look-ahead of the keyframe search, comma-separated key=value: depth, depth_max, side, tree (any shape
//                       gives the same keyframes: tests/test_gpu_adaptive.py)
//   ECAL_GRID_TOL_PX    the grid finder's hole tolerance (default: the reference's 20 px); ECAL_BO_BIG_ARENA: bytes of the member-order
//                       kernel's global arena (tests of its overflow path)
//   ECAL_MEDIAN_TIES, ECAL_TAIL_MODE, ECAL_ROCTX (ecal_init: the defaults of ecal_set_median_ties / ecal_set_tail_mode /
//   ecal_set_profile_ranges); ECAL_HOST_THREADS, ECAL_HOST_ARROW_PARTS (arrow_host_parts.hpp: the solver's host half)
void ecal_read_switches(ecal_switches &sw) {
    auto has = [](const char *var, const char *word) -> bool {      // `word` is an element of the comma-separated list in `var`
        const char *e = getenv(var);
        if (!e) return false;
        const size_t n = strlen(word);
        for (const char *p = e; *p;) {
            const char *q = strchr(p, ',');
            const size_t len = q ? (size_t) (q - p) : strlen(p);
            if (len == n && strncmp(p, word, n) == 0) return true;
            p += len + (q ? 1 : 0);
        }
        return false;
    };
    auto value = [](const char *var, const char *key, long long &out) -> bool {   // key=value in the comma-separated list
        const char *e = getenv(var);
        if (!e) return false;
        const size_t n = strlen(key);
        for (const char *p = e; *p;) {
            const char *q = strchr(p, ',');
            const size_t len = q ? (size_t) (q - p) : strlen(p);
            if (len > n + 1 && strncmp(p, key, n) == 0 && p[n] == '=') {
                out = atoll(p + n + 1);
                return true;
            }
            p += len + (q ? 1 : 0);
        }
        return false;
    };
    sw = ecal_switches();
    sw.slice_no_pixel = has("ECAL_FORCE", "slice_general");
    sw.dbscan_no_pixel = has("ECAL_FORCE", "dbscan_general");
    sw.dbscan_generic_disc = has("ECAL_FORCE", "dbscan_generic_disc");
    sw.extract_no_inline_ties = has("ECAL_FORCE", "extract_no_inline_ties");
    sw.bounds_two_kernels = has("ECAL_FORCE", "bounds_two_kernels");
    sw.grid_one_wave = has("ECAL_FORCE", "grid_one_wave");
    sw.grid_serial_walk = has("ECAL_FORCE", "grid_serial_walk");
    sw.solver_no_stream = has("ECAL_FORCE", "solver_no_stream");
    sw.latency_forms = has("ECAL_FORCE", "latency_forms") ? 2 : (has("ECAL_FORCE", "latency_two_pass") ? 1 : 0);
    sw.adaptive_trace = has("ECAL_TRACE", "adaptive");
    sw.grid_debug = has("ECAL_TRACE", "grid");
    sw.solver_trace = has("ECAL_TRACE", "solver");
    sw.load_trace = has("ECAL_TRACE", "load");
    long long v;
    if (value("ECAL_ADAPTIVE_SHAPE", "depth", v)) sw.adaptive_depth = (int) v;
    if (value("ECAL_ADAPTIVE_SHAPE", "depth_max", v)) sw.adaptive_depth_max = (int) v;
    if (value("ECAL_ADAPTIVE_SHAPE", "side", v)) sw.adaptive_side = (int) v;
    if (value("ECAL_ADAPTIVE_SHAPE", "tree", v)) sw.adaptive_tree = (int) v;
    if (const char *e = getenv("ECAL_BO_BIG_ARENA")) sw.bo_big_arena = (unsigned long long) atoll(e);
    if (const char *e = getenv("ECAL_GRID_TOL_PX")) sw.grid_tol_px = atof(e);
}

// tests: the switches again, after the environment changed under a live context
extern "C" int ecal_debug_reload_env(ecal_ctx *ctx) {
    if (!ctx) return ECAL_ERR_INVALID;
    ecal_read_switches(ctx->sw);
    return ECAL_OK;
}

uint32_t *ecal_zero_words(ecal_ctx *ctx, hipStream_t st, uint32_t n) {
    if (n > ZERO_RING_GRAIN) return nullptr;
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

unsigned char *ecal_fetch_pinned(ecal_ctx *ctx, size_t bytes) {
    if (ctx->fetch_pinned_cap >= bytes) return ctx->fetch_pinned;
    if (ctx->fetch_pinned) (void) hipHostFree(ctx->fetch_pinned);
    ctx->fetch_pinned = nullptr;
    ctx->fetch_pinned_cap = 0;
    const size_t want = (bytes + bytes / 4 + 4095) & ~(size_t) 4095;
    if (hipHostMalloc((void **) &ctx->fetch_pinned, want, hipHostMallocDefault) != hipSuccess) {
        (void) hipGetLastError();
        ctx->fetch_pinned = nullptr;
        return nullptr;
    }
    ctx->fetch_pinned_cap = want;
    return ctx->fetch_pinned;
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
    if (ctx->fetch_pinned) (void) hipHostFree(ctx->fetch_pinned);
    if (ctx->wb_done) (void) hipEventDestroy(ctx->wb_done);
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
