// In-library collective: an RCCL communicator per context, one rank per GPU / process.
//
// The reference has no distributed backend; BASELINE.json's north_star asks for the calibration views and
// the spline residuals sharded one batch per GPU with the per-view J^T J / J^T r blocks summed by an RCCL all-reduce over
// xGMI.  The messages are tiny (91 .. 170 doubles per evaluation): latency bound, one ncclAllReduce on the
// solver's own stream each, nothing proportional to the number of control points or views crosses the links.
// ecal_solver_solve and ecal_calibrate_views use this communicator only when the caller says so: options.allreduce =
// ecal_comm_allreduce, options.allreduce_user = the context (NULL always means a rank-local call; the callback seam takes
// any other transport, e.g. the gloo tests).
#include "ecal_ctx.hpp"

#include <rccl/rccl.h>

static_assert(sizeof(ncclUniqueId) == ECAL_COMM_ID_BYTES, "ECAL_COMM_ID_BYTES must be the size of ncclUniqueId");

#define ECAL_NCCL_TRY(ctx, call)                                                                     \
    do {                                                                                              \
        ncclResult_t r__ = (call);                                                                    \
        if (r__ != ncclSuccess) {                                                                     \
            (ctx)->last_error = std::string(#call) + ": " + ncclGetErrorString(r__);                  \
            return ECAL_ERR_COMM;                                                                     \
        }                                                                                             \
    } while (0)

extern "C" int ecal_comm_unique_id(void *id_out) {
    if (!id_out) return ECAL_ERR_INVALID;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return ECAL_ERR_COMM;
    memcpy(id_out, &id, sizeof(id));
    return ECAL_OK;
}

extern "C" int ecal_comm_init(ecal_ctx *ctx, const void *unique_id, int rank, int world_size) {
    if (!ctx || !unique_id || world_size < 1 || rank < 0 || rank >= world_size) return ECAL_ERR_INVALID;
    if (ctx->comm) {
        ctx->last_error = "the context already has a communicator";
        return ECAL_ERR_INVALID;
    }
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm = nullptr;
    ECAL_NCCL_TRY(ctx, ncclCommInitRank(&comm, world_size, id, rank));
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_size = world_size;
    return ECAL_OK;
}

extern "C" int ecal_comm_destroy(ecal_ctx *ctx) {
    if (!ctx) return ECAL_ERR_INVALID;
    if (ctx->comm) {
        (void) hipSetDevice(ctx->device);
        (void) ncclCommDestroy((ncclComm_t) ctx->comm);
        ctx->comm = nullptr;
    }
    ctx->comm_rank = 0;
    ctx->comm_size = 1;
    return ECAL_OK;
}

extern "C" int ecal_comm_size(const ecal_ctx *ctx) { return ctx ? ctx->comm_size : ECAL_ERR_INVALID; }
extern "C" int ecal_comm_rank(const ecal_ctx *ctx) { return ctx ? ctx->comm_rank : ECAL_ERR_INVALID; }

extern "C" int ecal_comm_allreduce_sum_dev(ecal_ctx *ctx, double *d_buf, size_t n_doubles, void *stream) {
    if (!ctx || (n_doubles && !d_buf)) return ECAL_ERR_INVALID;
    if (!ctx->comm || n_doubles == 0) return ECAL_OK;   // one rank: the sum is the buffer itself
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    ECAL_NCCL_TRY(ctx, ncclAllReduce(d_buf, d_buf, n_doubles, ncclDouble, ncclSum, (ncclComm_t) ctx->comm, (hipStream_t) stream));
    return ECAL_OK;
}

// the ecal_allreduce_fn a caller puts into its options to all-reduce through the context's communicator (user = the context)
extern "C" int ecal_comm_allreduce(void *user, double *d_buf, size_t n_doubles, void *stream) {
    ecal_ctx *ctx = (ecal_ctx *) user;
    if (ctx && !ctx->comm) {   // asked for a collective on a context that never joined one: say so instead of summing nothing
        ctx->last_error = "ecal_comm_allreduce: the context has no communicator (ecal_comm_init)";
        return ECAL_ERR_COMM;
    }
    return ecal_comm_allreduce_sum_dev(ctx, d_buf, n_doubles, stream);
}
