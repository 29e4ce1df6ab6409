// Multi-GPU start-up: the only collective of the whole job.  Rank 0 parses and packs
// the checkpoint; every rank then receives the packed arena (1.9 GB at f16) with ONE RCCL
// broadcast over xGMI.  The forward passes that follow share nothing between ranks.
#include <rccl/rccl.h>

#include <cstring>

#include "model.h"

using namespace me;

#define ME_NCCL(expr)                                                                        \
    do {                                                                                     \
        ncclResult_t r__ = (expr);                                                           \
        if (r__ != ncclSuccess)                                                              \
            ::me::fail(ME_ERR_RCCL, "%s: %s (%s:%d)", #expr, ncclGetErrorString(r__), __FILE__, \
                       __LINE__);                                                            \
    } while (0)

extern "C" {

int32_t me_rccl_unique_id(void* id128) {
    if (!id128) return ME_ERR_BAD_ARG;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return ME_ERR_RCCL;
    memcpy(id128, &id, sizeof id);
    return ME_OK;
}

int32_t me_bcast_weights(me_ctx* ctx, const void* id128, int32_t rank, int32_t nranks) {
    if (!ctx || !id128) return ME_ERR_BAD_ARG;
    ncclComm_t comm = nullptr;
    try {
        ME_CHECK(nranks >= 1 && rank >= 0 && rank < nranks, ME_ERR_BAD_ARG, "rank %d of %d", rank,
                 nranks);
        if (rank == 0)
            ME_CHECK(ctx->finalized, ME_ERR_NOT_READY, "rank 0 must finalize its weights first");
        ME_HIP(hipSetDevice(ctx->device));
        {
            // also with one rank: the communicator is created, the (in-place, root = self) broadcast runs and
            // the communicator is destroyed, so a single-GPU box exercises every RCCL call of this path
            ncclUniqueId id;
            memcpy(&id, id128, sizeof id);
            ME_NCCL(ncclCommInitRank(&comm, nranks, id, rank));
            ME_NCCL(ncclBroadcast(ctx->arena, ctx->arena, ctx->arena_bytes, ncclUint8, 0, comm,
                                  ctx->stream));
            ME_HIP(hipStreamSynchronize(ctx->stream));
            ME_NCCL(ncclCommDestroy(comm));
            comm = nullptr;
        }
        for (WeightSlot& s : ctx->slots) s.loaded = true;
        build_fp8_weights(ctx);
        ctx->finalized = true;
        ctx->drop_graph(), ++ctx->weights_generation;
    } catch (const me::Error& e) {
        if (comm) (void)ncclCommAbort(comm);
        ctx->last_error = e.msg;
        return e.code;
    }
    return ME_OK;
}

}  // extern "C"
