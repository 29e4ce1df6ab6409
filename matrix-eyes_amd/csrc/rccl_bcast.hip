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
            // every rank must lay the arena out as rank 0 does (dtype, geometry, split mask): 16 bytes first, {arena
            // bytes, layout hash} of rank 0, compared before the payload moves -- ranks that disagree would otherwise
            // hang in the collective or adopt a mis-laid-out arena
            uint64_t mine[2] = {(uint64_t)ctx->arena_bytes, ctx->arena_layout_hash()}, root[2] = {0, 0};
            uint64_t* hdr = (uint64_t*)site_buf(ctx, "bcast.header", sizeof mine + 8);
            ME_HIP(hipMemcpyAsync(hdr, mine, sizeof mine, hipMemcpyHostToDevice, ctx->stream));
            ME_NCCL(ncclBroadcast(hdr, hdr, sizeof mine, ncclUint8, 0, comm, ctx->stream));
            ME_HIP(hipMemcpyAsync(root, hdr, sizeof root, hipMemcpyDeviceToHost, ctx->stream));
            ME_HIP(hipStreamSynchronize(ctx->stream));
            // the verdict is shared (a MIN all-reduce of one word), so that EVERY rank leaves before the payload
            // broadcast when one of them disagrees, rank 0 included
            const bool same = root[0] == mine[0] && root[1] == mine[1];
            uint32_t ok = same ? 1u : 0u;
            uint32_t* okd = (uint32_t*)(hdr + 2);
            ME_HIP(hipMemcpyAsync(okd, &ok, 4, hipMemcpyHostToDevice, ctx->stream));
            ME_NCCL(ncclAllReduce(okd, okd, 1, ncclUint32, ncclMin, comm, ctx->stream));
            ME_HIP(hipMemcpyAsync(&ok, okd, 4, hipMemcpyDeviceToHost, ctx->stream));
            ME_HIP(hipStreamSynchronize(ctx->stream));
            ME_CHECK(same, ME_ERR_BAD_ARG,
                     "me_bcast_weights: rank %d lays its weight arena out differently from rank 0 (%llu bytes, hash %016llx "
                     "against %llu, %016llx): dtype, me_model_config and split_operands must match on every rank",
                     rank, (unsigned long long)mine[0], (unsigned long long)mine[1], (unsigned long long)root[0],
                     (unsigned long long)root[1]);
            ME_CHECK(ok == 1u, ME_ERR_BAD_ARG,
                     "me_bcast_weights: another rank lays its weight arena out differently from rank 0 (dtype, "
                     "me_model_config and split_operands must match on every rank)");
            ME_NCCL(ncclBroadcast(ctx->arena, ctx->arena, ctx->arena_bytes, ncclUint8, 0, comm,
                                  ctx->stream));
            ME_HIP(hipStreamSynchronize(ctx->stream));
            ME_NCCL(ncclCommDestroy(comm));
            comm = nullptr;
        }
        for (WeightSlot& s : ctx->slots) s.loaded = true;
        ctx->factor_keep.clear();
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
