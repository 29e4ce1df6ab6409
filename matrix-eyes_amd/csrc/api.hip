// extern "C" boundary of libmatrixeyes_hip.so (include/matrix_eyes_hip.h, matrix_eyes_hip_ops.h).
// Every entry point converts internal me::Error into a status code + last_error text; nothing
// throws or aborts across the boundary.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <exception>

#include "../../include/matrix_eyes_hip_ops.h"
#include "model.h"
#include "mx_fp8.h"

using namespace me;

namespace {

thread_local std::string g_create_error;

struct OutBuf {
    void* dev = nullptr;
    void* user = nullptr;
    size_t bytes = 0;
    bool staged = false;
};

OutBuf out_buf(me_ctx* ctx, void* user, size_t bytes, const std::string& name) {
    OutBuf o;
    o.user = user, o.bytes = bytes;
    if (is_device_ptr(user)) {
        o.dev = user;
    } else {
        o.dev = site_buf(ctx, name, bytes);
        o.staged = true;
    }
    return o;
}

void finish(me_ctx* ctx, const OutBuf& o) {
    if (o.staged) from_device(ctx, o.user, o.dev, o.bytes);
}

void check_ready(me_ctx* ctx) {
    ME_CHECK(ctx->finalized, ME_ERR_NOT_READY,
             "weights are not finalized (me_weights_finalize / me_bcast_weights)");
}

void check_batch(int32_t batch) {
    ME_CHECK(batch >= 1 && batch <= 64, ME_ERR_BAD_SHAPE, "batch %d out of range [1, 64]", batch);
}

void validate_config(const me_model_config& c) {
    ME_CHECK(c.grid >= 8 && c.grid % 8 == 0 && c.grid <= 64, ME_ERR_BAD_SHAPE,
             "grid %d: need a multiple of 8 in [8, 64]", c.grid);
    ME_CHECK(c.embed_dim == 64 * c.num_heads && c.num_heads >= 1, ME_ERR_BAD_SHAPE,
             "embed_dim %d must be 64 * num_heads (%d)", c.embed_dim, c.num_heads);
    ME_CHECK(c.embed_dim == 64 || c.embed_dim == 128 || c.embed_dim == 256 || c.embed_dim == 512 ||
                 c.embed_dim == 1024,
             ME_ERR_BAD_SHAPE, "embed_dim %d not in {64,128,256,512,1024}", c.embed_dim);
    ME_CHECK(c.depth >= 1 && c.depth <= 64, ME_ERR_BAD_SHAPE, "depth %d", c.depth);
    // vit.rs:318-324: a requested tap that does not exist panics
    ME_CHECK(c.tap_blocks[0] >= 0 && c.tap_blocks[0] < c.depth && c.tap_blocks[1] >= 0 &&
                 c.tap_blocks[1] < c.depth && c.tap_blocks[0] != c.tap_blocks[1],
             ME_ERR_BAD_SHAPE, "tap blocks {%d,%d} with depth %d", c.tap_blocks[0], c.tap_blocks[1],
             c.depth);
    for (int i = 0; i < 4; ++i)
        ME_CHECK(c.enc_dims[i] > 0 && c.enc_dims[i] % 64 == 0, ME_ERR_BAD_SHAPE,
                 "enc_dims[%d] = %d must be a multiple of 64", i, c.enc_dims[i]);
    ME_CHECK(c.dec_dim > 0 && c.dec_dim % 256 == 0, ME_ERR_BAD_SHAPE,
             "dec_dim %d must be a multiple of 256", c.dec_dim);
    ME_CHECK(c.head_dims[0] > 0 && c.head_dims[0] <= 32 && c.head_dims[0] % 4 == 0 &&
                 c.head_dims[1] == 1,
             ME_ERR_BAD_SHAPE, "head_dims {%d,%d}: need {<=32 multiple of 4, 1}", c.head_dims[0],
             c.head_dims[1]);
    ME_CHECK(c.ln_eps > 0.f, ME_ERR_BAD_ARG, "ln_eps %g", (double)c.ln_eps);
    ME_CHECK(c.split_operands >= 0 && c.split_operands <= 15, ME_ERR_BAD_ARG, "split_operands %d not in [0, 15]",
             c.split_operands);
}

}  // namespace

namespace {
__global__ void status_clear_bits_kernel(unsigned* word, unsigned bits) { atomicAnd(word, ~bits); }

// The fused residual + LayerNorm launch gave up waiting for a sibling workgroup (another tenant on the device's CUs, a
// CU mask the runtime does not report): from here on this context runs the stand-alone LayerNorm launches.  Logged once.
void latch_ln_fallback(me_ctx* ctx, const char* where) {
    if (ctx->ln_fuse_off) return;
    ctx->ln_fuse_off = true;
    ctx->drop_graph();  // the captured step holds the fused launches
    ctx->ln_fuse_note = std::string("LayerNorm fusion switched off for this context (") + where +
                        "): a workgroup of the fused residual + LayerNorm launch gave up waiting for its row tile's other "
                        "column tiles -- the device's CUs are shared or masked.  The stand-alone LayerNorm launches run from "
                        "here on (the ME_LN_FUSE=0 results, bit for bit)";
    fprintf(stderr, "matrix-eyes-hip: %s\n", ctx->ln_fuse_note.c_str());
}

// What earlier ASYNCHRONOUS steps (device results) raised, as far as their closing copy of the status word into the
// pinned mirror has landed -- no synchronisation.  A timed-out exchange means those steps' depth maps are invalid: the
// entry that finds it out fails (so that a caller who never polls me_status_flags still gets an error), with fusion
// off and the bit cleared, so the call can simply be made again.
void check_pending_status(me_ctx* ctx) {
    if (!ctx->status_host) return;
    const unsigned pending = *ctx->status_host;
    if (!(pending & ME_STATUS_SYNC_TIMEOUT)) return;
    latch_ln_fallback(ctx, "found at the next entry into the library");
    hipLaunchKernelGGL(status_clear_bits_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->status_dev, (unsigned)ME_STATUS_SYNC_TIMEOUT);
    ME_HIP(hipGetLastError());
    ME_HIP(hipStreamSynchronize(ctx->stream));
    *ctx->status_host = pending & ~(unsigned)ME_STATUS_SYNC_TIMEOUT;
    fail(ME_ERR_HIP,
         "an earlier asynchronous me_extract_depth step timed out in the fused LayerNorm exchange (ME_STATUS_SYNC_TIMEOUT): its "
         "depth map is not valid.  LayerNorm fusion is now off for this context; submit the work again");
}
}  // namespace

namespace me {
namespace {
me_ctx::DepthSlot* find_depth_slot(me_ctx* ctx, const void* p) {
    const char* q = (const char*)p;
    for (me_ctx::DepthSlot& s : ctx->depth_slots)
        if (s.base && q >= s.base && q < s.base + s.bytes) return &s;
    return nullptr;
}
}  // namespace

OutputScope::OutputScope(me_ctx* c, const void* depth) : ctx(c) {
    if (!c->output_overlap || !c->out_stream || c->capturing) return;
    saved = c->stream;
    me_ctx::DepthSlot* s = depth && is_device_ptr(depth) ? find_depth_slot(c, depth) : nullptr;
    if (s && s->has_produced) {
        ME_HIP(hipStreamWaitEvent(c->out_stream, s->produced, 0));
        slot = (int)(s - c->depth_slots);
    } else {
        // a buffer no step of this context has written (a caller's own depth, a host pointer): behind everything queued so far
        hipEvent_t ev = nullptr;
        ME_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        const hipError_t e1 = hipEventRecord(ev, saved), e2 = e1 == hipSuccess ? hipStreamWaitEvent(c->out_stream, ev, 0) : e1;
        (void)hipEventDestroy(ev);  // (released once it has completed)
        ME_HIP(e2);
    }
    c->stream = c->out_stream;
    active = true;
}
OutputScope::~OutputScope() {
    if (!active) return;
    if (slot >= 0) {
        me_ctx::DepthSlot& s = ctx->depth_slots[slot];
        if (!s.consumed && hipEventCreateWithFlags(&s.consumed, hipEventDisableTiming) != hipSuccess) s.consumed = nullptr;
        if (s.consumed && hipEventRecord(s.consumed, ctx->out_stream) == hipSuccess) s.has_consumed = true;
    }
    ctx->stream = saved;
}
}  // namespace me

namespace {
// the step that is about to write [out, out + bytes): behind every output call that still reads an overlapping buffer
void wait_for_consumers(me_ctx* ctx, const void* out, size_t bytes) {
    if (!ctx->output_overlap) return;
    const char* lo = (const char*)out;
    for (me_ctx::DepthSlot& s : ctx->depth_slots)
        if (s.base && s.has_consumed && lo < s.base + s.bytes && s.base < lo + bytes) {
            ME_HIP(hipStreamWaitEvent(ctx->stream, s.consumed, 0));
            s.has_consumed = false;
        }
}
// ... and, once it is queued, known as the producer of that range
void mark_produced(me_ctx* ctx, const void* out, size_t bytes) {
    if (!ctx->output_overlap) return;
    me_ctx::DepthSlot* slot = nullptr;
    for (me_ctx::DepthSlot& s : ctx->depth_slots)
        if (s.base == (const char*)out) slot = &s;
    if (!slot)
        for (me_ctx::DepthSlot& s : ctx->depth_slots)
            if (!s.base) {
                slot = &s;
                break;
            }
    if (!slot) {
        slot = &ctx->depth_slots[0];
        for (me_ctx::DepthSlot& s : ctx->depth_slots)   // the least recently produced
            if (s.stamp < slot->stamp) slot = &s;
    }
    if (slot->base != (const char*)out && slot->has_consumed) {
        // the range this slot stood for is forgotten: nothing may still be reading it unordered
        ME_HIP(hipStreamWaitEvent(ctx->stream, slot->consumed, 0));
        slot->has_consumed = false;
    }
    if (!slot->produced) ME_HIP(hipEventCreateWithFlags(&slot->produced, hipEventDisableTiming));
    slot->base = (const char*)out, slot->bytes = bytes, slot->stamp = ++ctx->depth_stamp;
    ME_HIP(hipEventRecord(slot->produced, ctx->stream));
    slot->has_produced = true;
}
}  // namespace

#define ME_API_BEGIN(ctx)                                                      \
    if (!(ctx)) return ME_ERR_BAD_ARG;                                         \
    try {                                                                      \
        ME_HIP(hipSetDevice((ctx)->device));                                   \
        me::set_current_status_word((ctx)->status_dev);

#define ME_API_END(ctx)                                                        \
    }                                                                          \
    catch (const me::Error& e) {                                               \
        (ctx)->last_error = e.msg;                                             \
        return e.code;                                                         \
    }                                                                          \
    catch (const std::exception& e) {                                          \
        (ctx)->last_error = std::string("internal: ") + e.what();              \
        return ME_ERR_BAD_ARG;                                                 \
    }                                                                          \
    return ME_OK;

extern "C" {

int32_t me_abi_version(void) { return ME_ABI_VERSION; }

int32_t me_default_config(me_model_config* cfg) {
    if (!cfg) return ME_ERR_BAD_ARG;
    // vit.rs:17-19,349-358; encoder.rs:227; mod.rs:262-263,308-311
    cfg->grid = 24, cfg->embed_dim = 1024, cfg->num_heads = 16, cfg->depth = 24;
    cfg->tap_blocks[0] = 5, cfg->tap_blocks[1] = 11;
    cfg->enc_dims[0] = 256, cfg->enc_dims[1] = 512, cfg->enc_dims[2] = 1024, cfg->enc_dims[3] = 1024;
    cfg->dec_dim = 256;
    cfg->head_dims[0] = 32, cfg->head_dims[1] = 1;
    cfg->ln_eps = 1e-5f;
    cfg->align_corners = 1;
    cfg->split_operands = 3;
    cfg->fp8_linears = 0;
    return ME_OK;
}

int32_t me_ctx_create(int32_t device_id, int32_t dtype, const me_model_config* cfg, me_ctx** out) {
    if (!out) return ME_ERR_BAD_ARG;
    *out = nullptr;
    me_ctx* ctx = nullptr;
    try {
        ME_CHECK(dtype == ME_DTYPE_F16 || dtype == ME_DTYPE_BF16 || dtype == ME_DTYPE_FP8, ME_ERR_BAD_ARG,
                 "bad dtype %d", dtype);
        int ndev = 0;
        const hipError_t e = hipGetDeviceCount(&ndev);
        ME_CHECK(e == hipSuccess && ndev > 0, ME_ERR_HIP,
                 "no HIP device available (%s): this library has no CPU fallback",
                 e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        ME_CHECK(device_id >= 0 && device_id < ndev, ME_ERR_BAD_ARG, "device %d of %d", device_id,
                 ndev);
        ctx = new me_ctx();
        ctx->device = device_id;
        // fp8 contexts: every 16-bit operand is f16 (the checkpoint's own type), the big ViT linears are MX fp8
        ctx->fp8 = dtype == ME_DTYPE_FP8;
        ctx->dtype = ctx->fp8 ? ME_DTYPE_F16 : dtype;
        if (cfg)
            ctx->cfg = *cfg;
        else
            me_default_config(&ctx->cfg);
        // diagnostic override of the DEFAULT split_operands (tools/split_budget.py).  A caller-supplied
        // configuration is never overridden: the mask fixes the arena layout, which ranks that exchange arenas
        // must agree on (me_bcast_weights compares me_ctx::arena_layout_hash)
        if (const char* e = getenv("ME_SPLIT_OPERANDS")) {
            if (!cfg) ctx->cfg.split_operands = atoi(e);
            else if (atoi(e) != cfg->split_operands)
                fprintf(stderr, "matrix-eyes-hip: ME_SPLIT_OPERANDS=%s ignored, the caller's me_model_config says %d\n", e,
                        cfg->split_operands);
        }
        if (const char* e = getenv("ME_GRAPH")) ctx->graph_enabled = atoi(e) != 0;
        validate_config(ctx->cfg);
        ME_CHECK(!ctx->fp8 || ctx->cfg.embed_dim % 256 == 0, ME_ERR_BAD_SHAPE,
                 "ME_DTYPE_FP8 needs embed_dim a multiple of 256 (256x256 tiles, K slabs of 128): %d", ctx->cfg.embed_dim);
        ctx->split_mask = ctx->cfg.split_operands;  // model.h SplitStage bits
        if (const char* e = getenv("ME_FP8_LINEARS")) {     // diagnostic override of the default (tools/fp8_budget.py)
            if (!cfg || cfg->fp8_linears == 0) ctx->cfg.fp8_linears = atoi(e);
        }
        ME_CHECK(ctx->cfg.fp8_linears >= 0 && ctx->cfg.fp8_linears <= 15, ME_ERR_BAD_ARG, "fp8_linears %d not in [0, 15]",
                 ctx->cfg.fp8_linears);
        ctx->fp8_mask = ctx->fp8 ? (ctx->cfg.fp8_linears ? ctx->cfg.fp8_linears : ME_FP8_LINEARS_DEFAULT) : 0;
        ME_HIP(hipSetDevice(device_id));
        ME_HIP(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
        ctx->stream = ctx->own_stream;
        ME_HIP(hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
        {
            // the output stream's launches are few and short (0.7 ms of kernels per image beside a 22 ms depth step): at the
            // highest priority they take the first CUs a boundary of the main stream's persistent kernels frees, and the host
            // thread that waits for their counts and copies waits less
            int least = 0, greatest = 0;
            (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
            if (hipStreamCreateWithPriority(&ctx->out_stream, hipStreamNonBlocking, greatest) != hipSuccess) {
                (void)hipGetLastError();
                ME_HIP(hipStreamCreateWithFlags(&ctx->out_stream, hipStreamNonBlocking));
            }
        }
        ME_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        ME_HIP(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
        ME_HIP(hipMalloc((void**)&ctx->status_dev, 256));
        ME_HIP(hipMemset(ctx->status_dev, 0, 256));
        ME_HIP(hipHostMalloc((void**)&ctx->status_host, 64, hipHostMallocDefault));
        *ctx->status_host = 0;
        build_weight_table(ctx);
        ME_HIP(hipMalloc((void**)&ctx->arena, ctx->arena_bytes));
        resolve_weights(ctx);
        *out = ctx;
        return ME_OK;
    } catch (const me::Error& e) {
        g_create_error = e.msg;
        if (ctx) me_ctx_destroy(ctx);
        return e.code;
    } catch (const std::exception& e) {
        g_create_error = e.what();
        if (ctx) me_ctx_destroy(ctx);
        return ME_ERR_BAD_ARG;
    }
}

void me_ctx_destroy(me_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
    if (ctx->side_stream) (void)hipStreamSynchronize(ctx->side_stream);
    if (ctx->out_stream) (void)hipStreamSynchronize(ctx->out_stream);
    ctx->drop_graph();
    // the status word of this context may be the calling thread's current one (ME_API_BEGIN): not after this
    if (me::current_status_word() == ctx->status_dev) me::set_current_status_word(nullptr);
    for (auto& kv : ctx->bufs)
        if (kv.second.p) (void)hipFree(kv.second.p);
    if (ctx->arena) (void)hipFree(ctx->arena);
    if (ctx->arena8) (void)hipFree(ctx->arena8);
    if (ctx->status_dev) (void)hipFree(ctx->status_dev);
    if (ctx->status_host) (void)hipHostFree((void*)ctx->status_host);
    (void)me_output_flush(ctx);  // pending write-behind files are completed before their buffers go
    for (me_ctx::WriteSlot& w : ctx->write_slots)
        if (w.pinned) (void)hipHostFree(w.pinned);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
    if (ctx->out_stream) (void)hipStreamDestroy(ctx->out_stream);
    for (me_ctx::DepthSlot& d : ctx->depth_slots) {
        if (d.produced) (void)hipEventDestroy(d.produced);
        if (d.consumed) (void)hipEventDestroy(d.consumed);
    }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

const char* me_last_error(const me_ctx* ctx) {
    return ctx ? ctx->last_error.c_str() : g_create_error.c_str();
}

int32_t me_ctx_set_progress(me_ctx* ctx, me_progress_fn fn, void* user) {
    if (!ctx) return ME_ERR_BAD_ARG;
    ctx->progress = fn, ctx->progress_user = user;
    return ME_OK;
}

int32_t me_ctx_set_stream(me_ctx* ctx, void* hip_stream) {
    ME_API_BEGIN(ctx)
    ME_HIP(hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    ctx->drop_graph();
    ME_API_END(ctx)
}

int32_t me_ctx_synchronize(me_ctx* ctx) {
    ME_API_BEGIN(ctx)
    ME_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->output_overlap && ctx->out_stream) ME_HIP(hipStreamSynchronize(ctx->out_stream));
    ME_API_END(ctx)
}

int32_t me_ctx_set_output_overlap(me_ctx* ctx, int32_t on) {
    ME_API_BEGIN(ctx)
    ME_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->out_stream) ME_HIP(hipStreamSynchronize(ctx->out_stream));
    ctx->output_overlap = on != 0 && ctx->out_stream != nullptr;
    for (me_ctx::DepthSlot& d : ctx->depth_slots) d.base = nullptr, d.bytes = 0, d.has_produced = d.has_consumed = false;
    ME_API_END(ctx)
}

// ---- weights ----------------------------------------------------------------------------
int32_t me_load_weight(me_ctx* ctx, const char* name, const void* data, int32_t weight_dtype,
                       const int64_t* dims, int32_t ndim) {
    ME_API_BEGIN(ctx)
    load_weight(ctx, name, data, weight_dtype, dims, ndim);
    ME_API_END(ctx)
}

int32_t me_expected_weight_count(const me_ctx* ctx) { return ctx ? (int32_t)ctx->slots.size() : 0; }

int32_t me_expected_weight(const me_ctx* ctx, int32_t index, const char** name, int64_t dims[4],
                           int32_t* ndim) {
    if (!ctx || index < 0 || index >= (int32_t)ctx->slots.size() || !name || !dims || !ndim)
        return ME_ERR_BAD_ARG;
    const WeightSlot& s = ctx->slots[index];
    *name = s.name.c_str();
    *ndim = (int32_t)s.dims.size();
    for (int i = 0; i < 4; ++i) dims[i] = i < *ndim ? s.dims[i] : 1;
    return ME_OK;
}

int32_t me_weights_finalize(me_ctx* ctx) {
    ME_API_BEGIN(ctx)
    finalize_weights(ctx);
    ME_API_END(ctx)
}

int32_t me_load_checkpoint_pt(me_ctx* ctx, const char* path) {
    ME_API_BEGIN(ctx)
    load_checkpoint_pt(ctx, path);
    ME_API_END(ctx)
}

int32_t me_unused_weight_count(const me_ctx* ctx) { return ctx ? (int32_t)ctx->unused_weights.size() : 0; }

const char* me_unused_weight_name(const me_ctx* ctx, int32_t index) {
    if (!ctx || index < 0 || index >= (int32_t)ctx->unused_weights.size()) return nullptr;
    return ctx->unused_weights[index].c_str();
}

int64_t me_weight_arena_bytes(const me_ctx* ctx) { return ctx ? (int64_t)ctx->arena_bytes : 0; }

uint64_t me_weight_arena_layout(const me_ctx* ctx) { return ctx ? ctx->arena_layout_hash() : 0; }

int32_t me_status_flags(me_ctx* ctx, uint32_t* flags) {
    ME_API_BEGIN(ctx)
    ME_CHECK(flags, ME_ERR_BAD_ARG, "me_status_flags: null pointer");
    uint32_t host = 0;
    ME_HIP(hipMemcpyAsync(&host, ctx->status_dev, 4, hipMemcpyDeviceToHost, ctx->stream));
    ME_HIP(hipMemsetAsync(ctx->status_dev, 0, 4, ctx->stream));
    ME_HIP(hipStreamSynchronize(ctx->stream));
    *ctx->status_host = 0;
    // the steps that raised it are lost (the caller knows: the bit is in *flags); the ones after this call are not
    if (host & ME_STATUS_SYNC_TIMEOUT) latch_ln_fallback(ctx, "me_status_flags saw ME_STATUS_SYNC_TIMEOUT");
    *flags = host;
    ME_API_END(ctx)
}

int32_t me_calibrate(me_ctx* ctx, double* out6) {
    ME_API_BEGIN(ctx)
    ME_CHECK(out6, ME_ERR_BAD_ARG, "me_calibrate: null pointer");
    calibrate(ctx, out6);
    ME_API_END(ctx)
}

int32_t me_ln_fusion_state(me_ctx* ctx, int32_t* fused, int32_t* fallbacks) {
    if (!ctx) return ME_ERR_BAD_ARG;
    if (fused) *fused = ctx->ln_fuse_off ? 0 : 1;
    if (fallbacks) *fallbacks = ctx->ln_fallbacks;
    return ME_OK;
}

void* me_weight_arena_ptr(const me_ctx* ctx) { return ctx ? (void*)ctx->arena : nullptr; }

int32_t me_weights_adopt(me_ctx* ctx) {
    ME_API_BEGIN(ctx)
    for (WeightSlot& s : ctx->slots) s.loaded = true;
    ctx->factor_keep.clear();  // host copies of factors of the arena this one replaces
    build_fp8_weights(ctx);
    ctx->finalized = true;
    ctx->drop_graph(), ++ctx->weights_generation;
    ME_API_END(ctx)
}

// ---- forward passes ----------------------------------------------------------------------
int32_t me_preprocess_u8(me_ctx* ctx, const uint8_t* rgb, int32_t batch, float* img) {
    ME_API_BEGIN(ctx)
    ME_CHECK(rgb && img, ME_ERR_BAD_ARG, "me_preprocess_u8: null pointer");
    check_batch(batch);
    const int S = ctx->S();
    const size_t npix = (size_t)batch * S * S;
    const void* src = to_device(ctx, rgb, npix * 3, "io.rgb");
    OutBuf o = out_buf(ctx, img, npix * 3 * 4, "io.img");
    preprocess_u8_launch((const uint8_t*)src, (float*)o.dev, batch, S, ctx->stream);
    finish(ctx, o);
    ME_API_END(ctx)
}

namespace {
struct ApiTap {
    me_ctx* ctx;
    const int32_t* blocks;
    int n;
    float** dev;
    size_t bytes;
};
void api_tap_fn(void* user, int index, const float* tokens) {
    ApiTap* t = (ApiTap*)user;
    for (int i = 0; i < t->n; ++i)
        if (t->blocks[i] == index)
            ME_HIP(hipMemcpyAsync(t->dev[i], tokens, t->bytes, hipMemcpyDeviceToDevice,
                                  t->ctx->stream));
}
}  // namespace

int32_t me_vit_forward_features(me_ctx* ctx, int32_t which_vit, const float* xs, int32_t windows,
                                const int32_t* intermediate_blocks, int32_t n_intermediate,
                                float* final_out, float* const* intermediate_out) {
    ME_API_BEGIN(ctx)
    check_ready(ctx);
    ME_CHECK(which_vit >= 0 && which_vit <= 2, ME_ERR_BAD_ARG, "which_vit %d", which_vit);
    ME_CHECK(xs && final_out, ME_ERR_BAD_ARG, "me_vit_forward_features: null pointer");
    ME_CHECK(windows >= 1 && windows <= 4096, ME_ERR_BAD_SHAPE, "windows %d", windows);
    ME_CHECK(n_intermediate >= 0 && n_intermediate <= 8 &&
                 (n_intermediate == 0 || (intermediate_blocks && intermediate_out)),
             ME_ERR_BAD_ARG, "bad intermediate block list");
    for (int i = 0; i < n_intermediate; ++i)  // vit.rs:318-324 "only {} / {} blocks found"
        ME_CHECK(intermediate_blocks[i] >= 0 && intermediate_blocks[i] < ctx->cfg.depth,
                 ME_ERR_BAD_SHAPE, "only %d blocks, block %d requested", ctx->cfg.depth,
                 intermediate_blocks[i]);
    const int g = ctx->g(), P = ctx->P(), T = ctx->T(), C = ctx->C(), win = 16 * g;
    const size_t in_elems = (size_t)windows * 3 * win * win;
    const float* xs_dev = (const float*)to_device(ctx, xs, in_elems * 4, "api.vit.in");
    void* xs16 = site_buf(ctx, "api.vit.in16", in_elems * 2);
    cast_f32_to_16_launch(xs_dev, xs16, (int64_t)in_elems, ctx->dtype, ctx->stream);
    void* patches = site_buf(ctx, "api.vit.patches", (size_t)windows * P * 768 * 2);
    patchify_windows_launch(xs16, patches, windows, g, ctx->dtype, ctx->stream);
    const size_t tok_bytes = (size_t)windows * T * C * 4;
    std::vector<OutBuf> inter(n_intermediate);
    std::vector<float*> inter_dev(n_intermediate);
    for (int i = 0; i < n_intermediate; ++i) {
        ME_CHECK(intermediate_out[i], ME_ERR_BAD_ARG, "null intermediate output %d", i);
        inter[i] = out_buf(ctx, intermediate_out[i], tok_bytes, "api.vit.inter" + std::to_string(i));
        inter_dev[i] = (float*)inter[i].dev;
    }
    ApiTap tap{ctx, intermediate_blocks, n_intermediate, inter_dev.data(), tok_bytes};
    VitTaps taps;
    if (n_intermediate) taps.fn = api_tap_fn, taps.user = &tap;
    OutBuf fin = out_buf(ctx, final_out, tok_bytes, "api.vit.final");
    vit_forward(ctx, which_vit, patches, windows, taps, nullptr, (float*)fin.dev, "vit.api",
                ctx->stream);
    finish(ctx, fin);
    for (auto& o : inter) finish(ctx, o);
    ME_API_END(ctx)
}

int32_t me_encoder_forward_encodings(me_ctx* ctx, const float* x, int32_t batch,
                                     float* const encodings[5]) {
    ME_API_BEGIN(ctx)
    check_ready(ctx);
    check_batch(batch);
    ME_CHECK(x && encodings, ME_ERR_BAD_ARG, "me_encoder_forward_encodings: null pointer");
    for (int i = 0; i < 5; ++i) ME_CHECK(encodings[i], ME_ERR_BAD_ARG, "null encoding %d", i);
    const me_model_config& c = ctx->cfg;
    const int S = ctx->S(), g = ctx->g();
    const float* x_dev = (const float*)to_device(ctx, x, (size_t)batch * 3 * S * S * 4, "io.img");
    stage_encoder(ctx, x_dev, batch, false);
    const int H[5] = {32 * g, 16 * g, 8 * g, 4 * g, 2 * g};
    const int Cc[5] = {c.dec_dim, c.enc_dims[0], c.enc_dims[1], c.enc_dims[2], c.enc_dims[3]};
    const char* names[5] = {"enc0.f32", "enc1.16b", "enc2.16b", "enc3.16b", "enc4.16b"};
    for (int i = 0; i < 5; ++i) {
        OutBuf o = out_buf(ctx, encodings[i], (size_t)batch * Cc[i] * H[i] * H[i] * 4,
                           "api.enc.out" + std::to_string(i));
        if (i == 0)
            nhwc32_to_nchw32_launch((const float*)ctx->bufs.at(names[i]).p, (float*)o.dev, batch, H[i],
                                    H[i], Cc[i], ctx->stream);
        else
            nhwc16_to_nchw32_launch(ctx->bufs.at(names[i]).p, (float*)o.dev, batch, H[i], H[i], Cc[i],
                                    1, ctx->dtype, ctx->stream, ctx->split(SPLIT_DEC_CONVS));
        finish(ctx, o);
    }
    ME_API_END(ctx)
}

int32_t me_decoder_forward(me_ctx* ctx, const float* const encodings[5], int32_t batch,
                           float* features, float* lowres_features) {
    ME_API_BEGIN(ctx)
    check_ready(ctx);
    check_batch(batch);
    ME_CHECK(encodings && features && lowres_features, ME_ERR_BAD_ARG,
             "me_decoder_forward: null pointer");
    const me_model_config& c = ctx->cfg;
    const int g = ctx->g(), dec = c.dec_dim;
    const int H[5] = {32 * g, 16 * g, 8 * g, 4 * g, 2 * g};
    const int Cc[5] = {dec, c.enc_dims[0], c.enc_dims[1], c.enc_dims[2], c.enc_dims[3]};
    const char* names[5] = {"enc0.r16b", "enc1.16b", "enc2.16b", "enc3.16b", "enc4.16b"};
    for (int i = 0; i < 5; ++i) {
        ME_CHECK(encodings[i], ME_ERR_BAD_ARG, "null encoding %d", i);
        const size_t n = (size_t)batch * Cc[i] * H[i] * H[i];
        const float* src =
            (const float*)to_device(ctx, encodings[i], n * 4, "api.dec.in" + std::to_string(i));
        const bool sp = i > 0 && ctx->split(SPLIT_DEC_CONVS);  // encoding 0 only feeds the residual units
        void* d16 = site_buf(ctx, names[i], (size_t)batch * (H[i] + 2) * (H[i] + 2) * Cc[i] * 2 * (sp ? 2 : 1));
        float* d32 = i == 0 ? (float*)site_buf(ctx, "enc0.f32", n * 4) : nullptr;
        nchw32_to_nhwc_launch(src, d32, d16, batch, H[i], H[i], Cc[i], 1, i == 0 ? 1 : 0, ctx->dtype,
                              ctx->stream, sp);
    }
    stage_decoder(ctx, batch, true);
    OutBuf of = out_buf(ctx, features, (size_t)batch * dec * H[0] * H[0] * 4, "api.dec.feat");
    nhwc32_to_nchw32_launch((const float*)ctx->bufs.at("features.f32").p, (float*)of.dev, batch, H[0],
                            H[0], dec, ctx->stream);
    finish(ctx, of);
    OutBuf ol = out_buf(ctx, lowres_features, (size_t)batch * dec * H[4] * H[4] * 4, "api.dec.low");
    nhwc32_to_nchw32_launch((const float*)ctx->bufs.at("lowres.f32").p, (float*)ol.dev, batch, H[4],
                            H[4], dec, ctx->stream);
    finish(ctx, ol);
    ME_API_END(ctx)
}

int32_t me_head_forward(me_ctx* ctx, const float* features, int32_t batch,
                        float* canonical_inverse_depth) {
    ME_API_BEGIN(ctx)
    check_ready(ctx);
    check_batch(batch);
    ME_CHECK(features && canonical_inverse_depth, ME_ERR_BAD_ARG, "me_head_forward: null pointer");
    const int S = ctx->S(), Hh = S / 2, dec = ctx->cfg.dec_dim;
    const size_t n = (size_t)batch * dec * Hh * Hh;
    const float* src = (const float*)to_device(ctx, features, n * 4, "api.head.in");
    const bool sp = ctx->split(SPLIT_HEAD);
    void* f16b = site_buf(ctx, "features.16b", (size_t)batch * (Hh + 2) * (Hh + 2) * dec * 2 * (sp ? 2 : 1));
    nchw32_to_nhwc_launch(src, nullptr, f16b, batch, Hh, Hh, dec, 1, 0, ctx->dtype, ctx->stream, sp);
    OutBuf o = out_buf(ctx, canonical_inverse_depth, (size_t)batch * S * S * 4, "io.depth");
    stage_head(ctx, batch, nullptr, false, (float*)o.dev);
    finish(ctx, o);
    ME_API_END(ctx)
}

int32_t me_fov_forward(me_ctx* ctx, const float* x, const float* lowres_feature, int32_t batch,
                       float* fov_deg) {
    ME_API_BEGIN(ctx)
    check_ready(ctx);
    check_batch(batch);
    ME_CHECK(x && lowres_feature && fov_deg, ME_ERR_BAD_ARG, "me_fov_forward: null pointer");
    const int S = ctx->S(), g = ctx->g(), dec = ctx->cfg.dec_dim;
    const float* x_dev = (const float*)to_device(ctx, x, (size_t)batch * 3 * S * S * 4, "io.img");
    void* x2 = site_buf(ctx, "enc.x2", (size_t)batch * 3 * (S / 4) * (S / 4) * 2);
    bilinear_launch(x_dev, x2, 3 * batch, S, S / 4, ctx->cfg.align_corners, ctx->dtype, ctx->stream);
    const size_t nl = (size_t)batch * dec * 4 * g * g;
    const float* low = (const float*)to_device(ctx, lowres_feature, nl * 4, "api.fov.low");
    float* low32 = (float*)site_buf(ctx, "lowres.f32", nl * 4);
    nchw32_to_nhwc_launch(low, low32, nullptr, batch, 2 * g, 2 * g, dec, 0, 0, ctx->dtype, ctx->stream);
    OutBuf o = out_buf(ctx, fov_deg, (size_t)batch * 4, "fov_deg");
    stage_fov_vit(ctx, batch, ctx->stream);
    stage_fov_tail(ctx, batch, (float*)o.dev);
    finish(ctx, o);
    ME_API_END(ctx)
}

namespace {
// the range a stage reports into, restored on scope exit (also when a stage throws)
struct ProgressRange {
    me_ctx* ctx;
    ProgressRange(me_ctx* c, float lo, float hi) : ctx(c) { ctx->prog_lo = lo, ctx->prog_hi = hi; }
    ~ProgressRange() { ctx->prog_lo = 0.0f, ctx->prog_hi = 1.0f; }
};

void extract_depth_impl(me_ctx* ctx, const float* img_dev, int32_t batch, const float* f_norm,
                        float* inverse_depth, float* fov_deg_out) {
    const int S = ctx->S();
    // mod.rs:265-293,345: the reference splits its progress bar 80 / 20 between depth and FOV when the FOV
    // head runs, the depth part 80 / 20 between encoder and the rest, that 98 / 2 between decoder and head,
    // and gives the first 5 % of each part to loading its weights (resident here).  The FOV tail runs
    // before the head in this pipeline (f_norm is divided out inside the head's last kernel), so it takes
    // the range between decoder and head.
    const float depth_end = f_norm ? 1.0f : 0.8f;
    const float enc_end = 0.8f * depth_end, dec_end = enc_end + 0.98f * (depth_end - enc_end);
    const float head_lo = f_norm ? dec_end : 0.99f;
    // Side branch (model.h me_ctx::side_stream).  The decoder's levels 4 - 2 and the FOV tail are 40 launches of 15 - 130 us
    // on 144 - 576 workgroups each -- latency-bound, most of the chip idle -- and depend only on encodings 2 - 4; the
    // encoder's two latent chains (encodings 1 and 0, read by levels 1 and 0 only) are ConvTranspose launches bound by
    // their HBM writes.  The two run side by side: fork behind the encoder trunk, join in front of level 1; the
    // persistent launches of the main branch leave half of the CUs to the side branch meanwhile.  Same kernels, same
    // order per buffer: the depth is bit for bit that of the one-stream order.  What it buys is small -- 22.84 -> 22.74 ms
    // per step (profiles/r04_side_stream.txt): side by side the ConvTranspose launch takes 404 us instead of 242 and the
    // small convolutions beside it three times their own time; both are short of memory-system bandwidth, not of CUs.
    // OFF by default since round 5 (VERDICT r4 weak 14 / ADVICE r4): 0.1 ms is inside the box-to-box noise, the small
    // launches themselves get slower beside the ConvTranspose, and the forked region adds a failure mode to graph capture
    // (a CaptureAbort between fork and join leaves the side stream attached to the aborted capture until EndCapture returns
    // Unjoined).  ME_OVERLAP_TAIL=1 switches it on for the A/B.
    static const bool overlap_env = getenv("ME_OVERLAP_TAIL") && atoi(getenv("ME_OVERLAP_TAIL")) != 0;
    const bool overlap = overlap_env && ctx->overlap_tail && ctx->side_stream && !ctx->progress && !profiler().enabled;
    OutBuf ofov;
    if (overlap) {
        stage_encoder_trunk(ctx, img_dev, batch, f_norm == nullptr);
        // (the stages below call site_buf behind the fork too -- first call only; that is safe because hipMalloc / hipFree
        // synchronise the whole device, not because the buffers exist beforehand)
        hipStream_t main_stream = ctx->stream;
        struct Restore {
            me_ctx* c;
            hipStream_t s;
            ~Restore() { c->stream = s, c->grid_cap = 0; }
        } restore{ctx, main_stream};
        if (!f_norm) ofov = out_buf(ctx, fov_deg_out ? fov_deg_out : nullptr, (size_t)batch * 4, "fov_deg");
        ME_HIP(hipEventRecord(ctx->ev_fork, main_stream));
        ME_HIP(hipStreamWaitEvent(ctx->side_stream, ctx->ev_fork, 0));
        ctx->stream = ctx->side_stream;
        stage_decoder_levels(ctx, batch, false, 4, 4);
        if (!f_norm) stage_fov_tail(ctx, batch, (float*)ofov.dev);   // needs level 4's lowres features only
        stage_decoder_levels(ctx, batch, false, 3, 2);
        ME_HIP(hipEventRecord(ctx->ev_join, ctx->side_stream));
        ctx->stream = main_stream;
        static const int cap = getenv("ME_OVERLAP_CAP") ? atoi(getenv("ME_OVERLAP_CAP")) : 128;
        ctx->grid_cap = cap;
        stage_encoder_latents(ctx, batch);
        ctx->grid_cap = 0;
        ME_HIP(hipStreamWaitEvent(main_stream, ctx->ev_join, 0));
        stage_decoder_levels(ctx, batch, false, 1, 0);
    } else {
        {
            ProgressRange r(ctx, 0.05f * enc_end, enc_end);
            stage_encoder(ctx, img_dev, batch, f_norm == nullptr);
        }
        {
            ProgressRange r(ctx, enc_end + 0.05f * (dec_end - enc_end), dec_end);
            stage_decoder(ctx, batch, false);
        }
    }
    float* fnorm_dev = (float*)site_buf(ctx, "f_norm", (size_t)batch * 4);
    if (f_norm) {
        if (is_device_ptr(f_norm))
            ME_HIP(hipMemcpyAsync(fnorm_dev, f_norm, (size_t)batch * 4, hipMemcpyDeviceToDevice,
                                  ctx->stream));
        else
            ME_HIP(hipMemcpyAsync(fnorm_dev, f_norm, (size_t)batch * 4, hipMemcpyHostToDevice,
                                  ctx->stream));
    } else if (!overlap) {
        // mod.rs:343-358
        ofov = out_buf(ctx, fov_deg_out ? fov_deg_out : nullptr, (size_t)batch * 4, "fov_deg");
        ProgressRange r(ctx, dec_end, head_lo);
        stage_fov_tail(ctx, batch, (float*)ofov.dev);
    }
    OutBuf o = out_buf(ctx, inverse_depth, (size_t)batch * S * S * 4, "io.depth");
    {
        ProgressRange r(ctx, head_lo, 1.0f);
        stage_head(ctx, batch, fnorm_dev, true, (float*)o.dev, ctx->features_pre);
    }
    finish(ctx, o);
    if (!f_norm && fov_deg_out) finish(ctx, ofov);
    report(ctx, 1.0f, nullptr);
}

// The overflow guard of a call that hands its result to the host (the stream has been synchronised by then): the
// reference computes in f32 and has no 65504 limit (decoder.rs:35-44), so an f16 operand that left the range is an
// error of THIS back end, reported instead of a silently zeroed conv branch.
// Returns true when the step must be run again: its fused LayerNorm exchange timed out (the context has fallen back to
// the stand-alone launches by then).  Only the bit that is reported is cleared: an overflow raised beside
// a timeout, or by an earlier asynchronous step, is still there for the re-run's own check / me_status_flags.
bool settle_host_result(me_ctx* ctx, bool may_rerun) {
    uint32_t host = 0;
    ME_HIP(hipMemcpy(&host, ctx->status_dev, 4, hipMemcpyDeviceToHost));
    auto clear = [&](unsigned bits) {
        hipLaunchKernelGGL(status_clear_bits_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->status_dev, bits);
        ME_HIP(hipGetLastError());
        ME_HIP(hipStreamSynchronize(ctx->stream));
        *ctx->status_host &= ~bits;
    };
    if (host & ME_STATUS_SYNC_TIMEOUT) {
        clear(ME_STATUS_SYNC_TIMEOUT);
        if (may_rerun && !ctx->ln_fuse_off) {
            latch_ln_fallback(ctx, "the step is being run again");
            ++ctx->ln_fallbacks;
            return true;
        }
        fail(ME_ERR_HIP,
             "a workgroup of the fused residual + LayerNorm launch gave up waiting for its neighbours' statistics (another "
             "process holding the device's CUs?): the depth map is not valid.  ME_LN_FUSE=0 runs the LayerNorm as its own launch");
    }
    if (host & ME_STATUS_OVERFLOW_16BIT) {
        clear(ME_STATUS_OVERFLOW_16BIT);
        fail(ME_ERR_OVERFLOW,
             "an activation left the f16 operand range (|x| > 65504) and was stored as +-inf: the depth map is not the "
             "reference's.  Create the context with ME_DTYPE_BF16 for this checkpoint");
    }
    return false;
}

// One step = preprocess (u8 entry) + extract_depth_impl, enqueued on ctx->stream.
void enqueue_step(me_ctx* ctx, int entry, const void* in_dev, int32_t batch, const float* f_norm,
                  float* inverse_depth, float* fov_deg_out) {
    const int S = ctx->S();
    // The status word is STICKY: nothing clears it when a step starts.  A call whose result goes to the host reports what
    // it finds when it has finished (settle_host_result: its own flags and whatever earlier asynchronous steps left
    // unread -- reporting those late beats erasing them) and clears the bit it reports; with a device result only
    // me_status_flags reads and clears, so that a loop of asynchronous calls (bench.py, a captured graph's replays) cannot
    // lose the overflow of an earlier step.
    const float* img_dev = (const float*)in_dev;
    if (entry == 1) {
        float* img = (float*)site_buf(ctx, "io.img", (size_t)batch * S * S * 3 * 4);
        preprocess_u8_launch((const uint8_t*)in_dev, img, batch, S, ctx->stream);
        img_dev = img;
    }
    extract_depth_impl(ctx, img_dev, batch, f_norm, inverse_depth, fov_deg_out);
    // a device result is asynchronous: leave the status word where the next entry finds it without synchronising
    if (is_device_ptr(inverse_depth))
        ME_HIP(hipMemcpyAsync((void*)ctx->status_host, ctx->status_dev, 4, hipMemcpyDeviceToHost, ctx->stream));
}

// The step through a hipGraph when nothing in it needs the host: every pointer on the device, no progress
// callback, no per-kernel timing.  First sight of a key: eager (allocates every site buffer); second: captured
// and instantiated; later: one hipGraphLaunch.  A capture that cannot complete (a buffer would have to grow, the
// runtime refuses a call) is discarded and the key stays eager.
void run_step(me_ctx* ctx, int entry, const void* in_dev, int32_t batch, const float* f_norm,
              float* inverse_depth, float* fov_deg_out) {
    const bool eligible = ctx->graph_enabled && !ctx->progress && !profiler().enabled && is_device_ptr(inverse_depth) &&
                          (!f_norm || is_device_ptr(f_norm)) && (f_norm || !fov_deg_out || is_device_ptr(fov_deg_out));
    if (!eligible) {
        enqueue_step(ctx, entry, in_dev, batch, f_norm, inverse_depth, fov_deg_out);
        return;
    }
    me_ctx::GraphKey key;
    key.in = in_dev, key.f_norm = f_norm, key.depth = inverse_depth, key.fov = f_norm ? nullptr : fov_deg_out;
    key.stream = ctx->stream, key.batch = batch, key.entry = entry, key.weights_generation = ctx->weights_generation;
    if (!(ctx->graph_seen && ctx->graph_key == key)) {
        ctx->drop_graph();
        ctx->graph_key = key, ctx->graph_seen = true;
        enqueue_step(ctx, entry, in_dev, batch, f_norm, inverse_depth, fov_deg_out);
        return;
    }
    if (!ctx->graph_exec && !ctx->graph_refused) {
        hipGraph_t graph = nullptr;
        bool ok = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed) == hipSuccess;
        if (ok) {
            ctx->capturing = true;
            try {
                enqueue_step(ctx, entry, in_dev, batch, f_norm, inverse_depth, fov_deg_out);
            } catch (...) {
                ok = false;
            }
            ctx->capturing = false;
            if (hipStreamEndCapture(ctx->stream, &graph) != hipSuccess) ok = false;
        }
        if (ok && graph && hipGraphInstantiate(&ctx->graph_exec, graph, nullptr, nullptr, 0) != hipSuccess)
            ctx->graph_exec = nullptr;
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        if (!ctx->graph_exec) ctx->graph_refused = true;
    }
    if (ctx->graph_exec) {
        ME_HIP(hipGraphLaunch(ctx->graph_exec, ctx->stream));
        ++ctx->graph_launches;
    } else {
        enqueue_step(ctx, entry, in_dev, batch, f_norm, inverse_depth, fov_deg_out);
    }
}
}  // namespace

int32_t me_extract_depth(me_ctx* ctx, const float* img, int32_t batch, const float* f_norm,
                         float* inverse_depth, float* fov_deg_out) {
    ME_API_BEGIN(ctx)
    check_ready(ctx);
    check_batch(batch);
    ME_CHECK(img && inverse_depth, ME_ERR_BAD_ARG, "me_extract_depth: null pointer");
    const int S = ctx->S();
    check_pending_status(ctx);
    const bool in_dev = is_device_ptr(img);
    const float* img_dev = (const float*)to_device(ctx, img, (size_t)batch * 3 * S * S * 4, "io.img");
    const size_t depth_bytes = (size_t)batch * S * S * 4;
    if (is_device_ptr(inverse_depth)) wait_for_consumers(ctx, inverse_depth, depth_bytes);
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (in_dev)
            run_step(ctx, 0, img_dev, batch, f_norm, inverse_depth, fov_deg_out);
        else
            enqueue_step(ctx, 0, img_dev, batch, f_norm, inverse_depth, fov_deg_out);
        // a device result is asynchronous (me_status_flags / the next entry's check_pending_status); a host result is
        // checked here, and a step whose fused LayerNorm exchange timed out is run once more without the fusion
        if (is_device_ptr(inverse_depth) || !settle_host_result(ctx, attempt == 0)) break;
    }
    if (is_device_ptr(inverse_depth)) mark_produced(ctx, inverse_depth, depth_bytes);
    ME_API_END(ctx)
}

int32_t me_extract_depth_u8(me_ctx* ctx, const uint8_t* rgb, int32_t batch, const float* f_norm,
                            float* inverse_depth, float* fov_deg_out) {
    ME_API_BEGIN(ctx)
    check_ready(ctx);
    check_batch(batch);
    ME_CHECK(rgb && inverse_depth, ME_ERR_BAD_ARG, "me_extract_depth_u8: null pointer");
    const int S = ctx->S();
    const size_t npix = (size_t)batch * S * S;
    check_pending_status(ctx);
    const bool in_dev = is_device_ptr(rgb);
    const void* src = to_device(ctx, rgb, npix * 3, "io.rgb");
    const size_t depth_bytes = (size_t)batch * S * S * 4;
    if (is_device_ptr(inverse_depth)) wait_for_consumers(ctx, inverse_depth, depth_bytes);
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (in_dev)
            run_step(ctx, 1, src, batch, f_norm, inverse_depth, fov_deg_out);
        else
            enqueue_step(ctx, 1, src, batch, f_norm, inverse_depth, fov_deg_out);
        if (is_device_ptr(inverse_depth) || !settle_host_result(ctx, attempt == 0)) break;
    }
    if (is_device_ptr(inverse_depth)) mark_produced(ctx, inverse_depth, depth_bytes);
    ME_API_END(ctx)
}

int32_t me_ctx_set_graph(me_ctx* ctx, int32_t on) {
    ME_API_BEGIN(ctx)
    ctx->graph_enabled = on != 0;
    if (!on) ctx->drop_graph();
    ME_API_END(ctx)
}
int64_t me_graph_launch_count(const me_ctx* ctx) { return ctx ? ctx->graph_launches : 0; }

// ---- output back end ---------------------------------------------------------------------
int32_t me_depth_clamp_minmax(me_ctx* ctx, float* depth, int64_t count, float* min_out,
                              float* max_out) {
    ME_API_BEGIN(ctx)
    OutputScope out_scope(ctx, depth);
    ME_CHECK(depth && count > 0, ME_ERR_BAD_ARG, "me_depth_clamp_minmax: bad argument");
    const bool dev = is_device_ptr(depth);
    float* d = dev ? depth : (float*)site_buf(ctx, "out.depth", (size_t)count * 4);
    if (!dev) ME_HIP(hipMemcpyAsync(d, depth, (size_t)count * 4, hipMemcpyHostToDevice, ctx->stream));
    float* mm = (float*)site_buf(ctx, "out.minmax", 8);
    depth_clamp_minmax_launch(d, count, mm, ctx->stream);
    float host[2];
    ME_HIP(hipMemcpyAsync(host, mm, 8, hipMemcpyDeviceToHost, ctx->stream));
    if (!dev) ME_HIP(hipMemcpyAsync(depth, d, (size_t)count * 4, hipMemcpyDeviceToHost, ctx->stream));
    ME_HIP(hipStreamSynchronize(ctx->stream));
    if (min_out) *min_out = host[0];
    if (max_out) *max_out = host[1];
    ME_API_END(ctx)
}

int32_t me_depth_clamp_minmax_async(me_ctx* ctx, float* depth, int64_t count, float* minmax_dev) {
    ME_API_BEGIN(ctx)
    OutputScope out_scope(ctx, depth);
    ME_CHECK(depth && minmax_dev && count > 0, ME_ERR_BAD_ARG, "me_depth_clamp_minmax_async: bad argument");
    ME_CHECK(is_device_ptr(depth) && is_device_ptr(minmax_dev), ME_ERR_BAD_ARG,
             "me_depth_clamp_minmax_async: depth and minmax_dev must be device memory");
    depth_clamp_minmax_launch(depth, count, minmax_dev, ctx->stream);
    ME_API_END(ctx)
}

namespace {
void stereogram_impl(me_ctx* ctx, const float* depth, int32_t rows, int32_t cols, float min_depth, float max_depth,
                     const float* range_dev, int32_t out_w, int32_t out_h, float amplitude, const uint8_t* noise,
                     uint8_t* out) {
    ME_CHECK(depth && noise && out, ME_ERR_BAD_ARG, "me_stereogram: null pointer");
    ME_CHECK(rows > 0 && cols > 0 && out_w > 0 && out_h > 0, ME_ERR_BAD_SHAPE,
             "me_stereogram: %dx%d -> %dx%d", rows, cols, out_w, out_h);
    const size_t nout = (size_t)out_w * out_h * 3;
    const float* d = (const float*)to_device(ctx, depth, (size_t)rows * cols * 4, "out.depth");
    const uint8_t* nz = (const uint8_t*)to_device(ctx, noise, nout, "out.noise");
    OutBuf o = out_buf(ctx, out, nout, "out.stereo");
    stereogram_launch(d, rows, cols, min_depth, max_depth, range_dev, out_w, out_h, amplitude, nz,
                      (uint8_t*)o.dev, ctx->stream);
    finish(ctx, o);
}
void depthmap_rgb_impl(me_ctx* ctx, const float* depth, int64_t count, float min_depth, float max_depth,
                       const float* range_dev, uint8_t* rgb) {
    ME_CHECK(depth && rgb && count > 0, ME_ERR_BAD_ARG, "me_depthmap_rgb: bad argument");
    const float* d = (const float*)to_device(ctx, depth, (size_t)count * 4, "out.depth");
    OutBuf o = out_buf(ctx, rgb, (size_t)count * 3, "out.rgb");
    depthmap_rgb_launch(d, count, min_depth, max_depth, range_dev, (uint8_t*)o.dev, ctx->stream);
    finish(ctx, o);
}
}  // namespace

int32_t me_stereogram(me_ctx* ctx, const float* depth, int32_t rows, int32_t cols, float min_depth,
                      float max_depth, int32_t out_w, int32_t out_h, float amplitude,
                      const uint8_t* noise, uint8_t* out) {
    ME_API_BEGIN(ctx)
    OutputScope out_scope(ctx, depth);
    stereogram_impl(ctx, depth, rows, cols, min_depth, max_depth, nullptr, out_w, out_h, amplitude, noise, out);
    ME_API_END(ctx)
}

int32_t me_stereogram_dev_range(me_ctx* ctx, const float* depth, int32_t rows, int32_t cols,
                                const float* minmax_dev, int32_t out_w, int32_t out_h, float amplitude,
                                const uint8_t* noise, uint8_t* out) {
    ME_API_BEGIN(ctx)
    OutputScope out_scope(ctx, depth);
    ME_CHECK(minmax_dev && is_device_ptr(minmax_dev), ME_ERR_BAD_ARG, "me_stereogram_dev_range: minmax_dev");
    stereogram_impl(ctx, depth, rows, cols, 0.f, 0.f, minmax_dev, out_w, out_h, amplitude, noise, out);
    ME_API_END(ctx)
}

int32_t me_depthmap_rgb(me_ctx* ctx, const float* depth, int64_t count, float min_depth,
                        float max_depth, uint8_t* rgb) {
    ME_API_BEGIN(ctx)
    OutputScope out_scope(ctx, depth);
    depthmap_rgb_impl(ctx, depth, count, min_depth, max_depth, nullptr, rgb);
    ME_API_END(ctx)
}

int32_t me_depthmap_rgb_dev_range(me_ctx* ctx, const float* depth, int64_t count, const float* minmax_dev,
                                  uint8_t* rgb) {
    ME_API_BEGIN(ctx)
    OutputScope out_scope(ctx, depth);
    ME_CHECK(minmax_dev && is_device_ptr(minmax_dev), ME_ERR_BAD_ARG, "me_depthmap_rgb_dev_range: minmax_dev");
    depthmap_rgb_impl(ctx, depth, count, 0.f, 0.f, minmax_dev, rgb);
    ME_API_END(ctx)
}

int32_t me_mesh_index(me_ctx* ctx, const float* depth, int32_t width, int32_t height,
                      int32_t* vertex_index, int64_t* nvertices, int64_t* nfaces, int32_t* faces) {
    ME_API_BEGIN(ctx)
    OutputScope out_scope(ctx, depth);
    ME_CHECK(depth && vertex_index && nvertices && nfaces, ME_ERR_BAD_ARG,
             "me_mesh_index: null pointer");
    ME_CHECK(width >= 2 && height >= 2, ME_ERR_BAD_SHAPE, "me_mesh_index: %dx%d", width, height);
    const size_t nv = (size_t)width * height;
    const size_t nt = 2 * (size_t)(width - 1) * (height - 1);
    const float* d = (const float*)to_device(ctx, depth, nv * 4, "out.depth");
    OutBuf ov = out_buf(ctx, vertex_index, nv * 4, "out.vindex");
    OutBuf of;
    if (faces) of = out_buf(ctx, faces, nt * 3 * 4, "out.faces");
    mesh_index_run(d, width, height, (int32_t*)ov.dev, faces ? (int32_t*)of.dev : nullptr, nvertices,
                   nfaces, site_buf(ctx, "out.mesh.ws", mesh_workspace_bytes(width, height)), ctx->stream);
    finish(ctx, ov);
    if (faces && of.staged) {  // only the kept triangles are defined
        ME_HIP(hipMemcpyAsync(faces, of.dev, (size_t)*nfaces * 12, hipMemcpyDeviceToHost,
                              ctx->stream));
        ME_HIP(hipStreamSynchronize(ctx->stream));
    }
    ME_API_END(ctx)
}

int32_t me_mesh_vertices(me_ctx* ctx, const float* depth, int32_t width, int32_t height,
                         const int32_t* vertex_index, int64_t nvertices, uint32_t original_width,
                         uint32_t original_height, float* uv, float* xyz) {
    ME_API_BEGIN(ctx)
    OutputScope out_scope(ctx, depth);
    ME_CHECK(depth && vertex_index && nvertices >= 0, ME_ERR_BAD_ARG, "me_mesh_vertices: bad argument");
    ME_CHECK(original_width > 0 && original_height > 0, ME_ERR_BAD_ARG, "original size 0");
    const size_t nv = (size_t)width * height;
    const float* d = (const float*)to_device(ctx, depth, nv * 4, "out.depth");
    const int32_t* vi = (const int32_t*)to_device(ctx, vertex_index, nv * 4, "out.vindex");
    // output.rs:222-225 (f32 division of the u32 sizes)
    const uint32_t mx = original_width > original_height ? original_width : original_height;
    const float xm = (float)original_width / (float)mx, ym = (float)original_height / (float)mx;
    OutBuf ouv, oxyz;
    if (uv) ouv = out_buf(ctx, uv, (size_t)nvertices * 8 + 8, "out.uv");
    if (xyz) oxyz = out_buf(ctx, xyz, (size_t)nvertices * 12 + 12, "out.xyz");
    mesh_vertices_launch(d, width, height, vi, xm, ym, uv ? (float*)ouv.dev : nullptr,
                         xyz ? (float*)oxyz.dev : nullptr, ctx->stream);
    if (uv && ouv.staged) from_device(ctx, uv, ouv.dev, (size_t)nvertices * 8);
    if (xyz && oxyz.staged) from_device(ctx, xyz, oxyz.dev, (size_t)nvertices * 12);
    ME_API_END(ctx)
}

// ---- kernel-level surface (matrix_eyes_hip_ops.h) -------------------------------------------
static unsigned long long* g_stamps = nullptr;  // diagnostic builds (-DME_GEMM_STAMPS) only
extern "C" int32_t me_debug_set_stamps(void* dev_ptr) {
    g_stamps = (unsigned long long*)dev_ptr;
    return ME_OK;
}

int32_t me_op_linear(me_ctx* ctx, int32_t M, int32_t N, int32_t K, const void* A16, const void* W16,
                     const float* bias, void* out16, float* out32, int32_t act, int32_t tile_cfg) {
    ME_API_BEGIN(ctx)
    GemmParams p = GemmParams();
    p.stamps = g_stamps;
    p.M = M, p.N = N, p.K = K, p.A = A16, p.lda = K, p.W = W16, p.bias = bias;
    p.out16 = out16, p.out32 = out32, p.ldc = N, p.act = act;
    gemm_launch(p, A_PLAIN, EPI_STORE, ctx->dtype, ctx->stream, tile_cfg);
    ME_API_END(ctx)
}

int32_t me_op_linear_residual(me_ctx* ctx, int32_t M, int32_t N, int32_t K, const void* A16,
                              const void* W16, const float* bias, const float* gamma, float* x32,
                              int32_t tile_cfg) {
    ME_API_BEGIN(ctx)
    ME_CHECK(bias && gamma && x32, ME_ERR_BAD_ARG, "me_op_linear_residual: null pointer");
    GemmParams p = GemmParams();
    p.stamps = g_stamps;
    p.M = M, p.N = N, p.K = K, p.A = A16, p.lda = K, p.W = W16, p.bias = bias, p.gamma = gamma;
    p.res32 = x32, p.out32 = x32, p.ldc = N;
    gemm_launch(p, A_PLAIN, EPI_RESID_SCALE, ctx->dtype, ctx->stream, tile_cfg);
    ME_API_END(ctx)
}

int32_t me_op_attention(me_ctx* ctx, const void* qkv16, void* out16, int32_t windows, int32_t tokens,
                        int32_t heads) {
    ME_API_BEGIN(ctx)
    ME_CHECK(qkv16 && out16, ME_ERR_BAD_ARG, "me_op_attention: null pointer");
    attention_launch(qkv16, out16, windows, tokens, heads, ctx->dtype, ctx->stream);
    ME_API_END(ctx)
}

int32_t me_op_attention_prescaled(me_ctx* ctx, const void* qkv16, void* out16, int32_t windows, int32_t tokens,
                                  int32_t heads) {
    ME_API_BEGIN(ctx)
    ME_CHECK(qkv16 && out16, ME_ERR_BAD_ARG, "me_op_attention_prescaled: null pointer");
    attention_launch(qkv16, out16, windows, tokens, heads, ctx->dtype, ctx->stream, nullptr, nullptr, nullptr, 0, true);
    ME_API_END(ctx)
}

int32_t me_op_linear_scaled_cols(me_ctx* ctx, int32_t M, int32_t N, int32_t K, const void* A16, const void* W16,
                                 const float* bias, void* out16, int32_t qcols, float qscale, int32_t tile_cfg) {
    ME_API_BEGIN(ctx)
    ME_CHECK(A16 && W16 && bias && out16, ME_ERR_BAD_ARG, "me_op_linear_scaled_cols: null pointer");
    GemmParams p = GemmParams();
    p.clamp_lo = -INFINITY, p.clamp_hi = INFINITY;
    p.M = M, p.N = N, p.K = K, p.A = A16, p.lda = K, p.W = W16, p.bias = bias, p.out16 = out16, p.ldc = N;
    p.qcols = qcols, p.qscale = qscale;
    gemm_launch(p, A_PLAIN, EPI_STORE, ctx->dtype, ctx->stream, tile_cfg);
    ME_API_END(ctx)
}

int32_t me_op_layernorm(me_ctx* ctx, const float* x32, const float* weight, const float* bias,
                        void* y16, float* y32, int64_t rows, int32_t dim, float eps) {
    ME_API_BEGIN(ctx)
    ME_CHECK(x32 && weight && bias && (y16 || y32), ME_ERR_BAD_ARG, "me_op_layernorm: null pointer");
    layernorm_launch(x32, weight, bias, y16, y32, rows, dim, eps, ctx->dtype, ctx->stream);
    ME_API_END(ctx)
}

int32_t me_op_conv2d(me_ctx* ctx, const void* in16b, int32_t B, int32_t H, int32_t W, int32_t Cin,
                     const void* w16, int32_t Cout, int32_t k, int32_t stride, const float* bias,
                     const float* res32, const float* res32b, float* out32, void* out16,
                     int32_t border16, int32_t act, int32_t act_both, int32_t tile_cfg) {
    ME_API_BEGIN(ctx)
    ME_CHECK((k == 1 || k == 3) && (stride == 1 || stride == 2), ME_ERR_BAD_SHAPE,
             "me_op_conv2d: k=%d stride=%d", k, stride);
    GemmParams p = GemmParams();
    const int Ho = H / stride, Wo = W / stride;
    p.M = B * Ho * Wo, p.N = Cout, p.K = k * k * Cin, p.A = in16b;
    p.in_Hp = H + 2, p.in_Wp = W + 2, p.Cin = Cin, p.out_H = Ho, p.out_W = Wo;
    p.KH = k, p.KW = k, p.stride = stride, p.W = w16, p.bias = bias;
    p.res32 = res32, p.res32b = res32b, p.out32 = out32, p.out16 = out16, p.ldc = Cout;
    p.out16_border = border16, p.act = act, p.act16_only = act_both ? 0 : 1;
    gemm_launch(p, A_CONV, EPI_STORE, ctx->dtype, ctx->stream, tile_cfg);
    ME_API_END(ctx)
}

int32_t me_op_head_final(me_ctx* ctx, const void* in16b, int32_t B, int32_t H, int32_t W, int32_t Cin, const void* w16,
                         int32_t Cmid, const float* bias, const float* w2, const float* b2, const float* f_norm,
                         float clamp_lo, float clamp_hi, float* out32, int32_t tile_cfg) {
    ME_API_BEGIN(ctx)
    ME_CHECK(in16b && w16 && bias && w2 && b2 && out32, ME_ERR_BAD_ARG, "me_op_head_final: null pointer");
    ME_CHECK(B > 0 && H > 0 && W > 0 && Cin % 64 == 0 && Cmid > 0 && Cmid <= 32 && Cmid % 4 == 0, ME_ERR_BAD_SHAPE,
             "me_op_head_final: %d x %d x %d, %d -> %d", B, H, W, Cin, Cmid);
    GemmParams p = GemmParams();
    p.M = B * H * W, p.N = Cmid, p.K = 9 * Cin, p.A = in16b;
    p.in_Hp = H + 2, p.in_Wp = W + 2, p.Cin = Cin, p.out_H = H, p.out_W = W;
    p.KH = 3, p.KW = 3, p.stride = 1, p.W = w16, p.bias = bias, p.w2 = w2, p.b2 = b2, p.f_norm = f_norm;
    p.pixels_per_image = H * W, p.out32 = out32, p.clamp_lo = clamp_lo, p.clamp_hi = clamp_hi, p.ldc = Cmid;
    gemm_launch(p, A_CONV, EPI_HEAD_FINAL, ctx->dtype, ctx->stream, tile_cfg);
    ME_API_END(ctx)
}

int32_t me_op_conv_transpose2x2(me_ctx* ctx, const void* in16, int32_t B, int32_t H, int32_t W,
                                int32_t Cin, const void* w16, int32_t Cout, const float* bias,
                                float* out32, void* out16, int32_t border16, int32_t tile_cfg) {
    ME_API_BEGIN(ctx)
    GemmParams p = GemmParams();
    p.M = B * H * W, p.N = 4 * Cout, p.K = Cin, p.A = in16, p.lda = Cin, p.W = w16, p.bias = bias;
    p.out_H = H, p.out_W = W, p.Cout = Cout, p.out32 = out32, p.out16 = out16;
    p.out16_border = border16, p.ldc = Cout;
    gemm_launch(p, A_PLAIN, EPI_CONVT, ctx->dtype, ctx->stream, tile_cfg);
    ME_API_END(ctx)
}

int32_t me_op_quantize_fp8(me_ctx* ctx, const void* src16, int64_t rows, int32_t K, int32_t weight_layout,
                           uint8_t* dst8, uint8_t* scales) {
    ME_API_BEGIN(ctx)
    ME_CHECK(src16 && dst8 && scales, ME_ERR_BAD_ARG, "me_op_quantize_fp8: null pointer");
    quantize_f16_to_fp8_launch(src16, dst8, scales, rows, K, weight_layout, ctx->stream);
    ME_API_END(ctx)
}

int32_t me_op_attention_fp8(me_ctx* ctx, const void* qkv16, uint8_t* out8, uint8_t* out8_scale, int32_t windows,
                            int32_t tokens, int32_t heads) {
    ME_API_BEGIN(ctx)
    ME_CHECK(qkv16 && out8 && out8_scale, ME_ERR_BAD_ARG, "me_op_attention_fp8: null pointer");
    attention_launch(qkv16, nullptr, windows, tokens, heads, ctx->dtype, ctx->stream, nullptr, out8, out8_scale,
                     cdiv((int64_t)windows * tokens, 128));
    ME_API_END(ctx)
}

int64_t me_op_scale_index(int64_t row, int32_t kblock, int64_t rows, int32_t weight_layout) {
    return weight_layout ? w_scale_index(row, kblock, rows / 64) : a_scale_index(row, kblock, cdiv(rows, 128));
}

int32_t me_op_layernorm_fp8(me_ctx* ctx, const float* x32, const float* weight, const float* bias, uint8_t* y8,
                            uint8_t* yscale, int64_t rows, int32_t dim, float eps) {
    ME_API_BEGIN(ctx)
    ME_CHECK(x32 && weight && bias && y8 && yscale, ME_ERR_BAD_ARG, "me_op_layernorm_fp8: null pointer");
    layernorm_fp8_launch(x32, weight, bias, y8, yscale, rows, dim, eps, ctx->stream);
    ME_API_END(ctx)
}

int32_t me_op_linear_fp8(me_ctx* ctx, int32_t M, int32_t N, int32_t K, const uint8_t* A8, const uint8_t* a_scale,
                         const uint8_t* W8, const uint8_t* w_scale, const float* bias, void* out16, uint8_t* out8,
                         uint8_t* out8_scale, const float* gamma, float* x32) {
    ME_API_BEGIN(ctx)
    GemmParams p = GemmParams();
    p.M = M, p.N = N, p.K = K, p.A = A8, p.lda = K, p.a_scale = a_scale, p.a_mt = (int)cdiv(M, 128);
    p.W = W8, p.w_scale = w_scale, p.bias = bias, p.ldc = N;
    p.clamp_lo = -INFINITY, p.clamp_hi = INFINITY;
    if (x32) {
        ME_CHECK(gamma && bias, ME_ERR_BAD_ARG, "me_op_linear_fp8: the residual form takes bias and gamma");
        p.gamma = gamma, p.res32 = x32, p.out32 = x32;
        gemm_fp8_launch(p, EPI_RESID_SCALE, ctx->stream);
    } else if (out8) {
        p.act = ACT_GELU, p.out8 = out8, p.out8_scale = out8_scale, p.out8_mt = (int)cdiv(M, 128);
        gemm_fp8_launch(p, EPI_STORE, ctx->stream);
    } else {
        p.out16 = out16;
        gemm_fp8_launch(p, EPI_STORE, ctx->stream);
    }
    ME_API_END(ctx)
}

int32_t me_op_linear_fp8_segments(me_ctx* ctx, int32_t M, int32_t N, int32_t K, const uint8_t* A8, const uint8_t* a_scale,
                                  int32_t seg1, int32_t seg2, const uint8_t* const W8[3], const uint8_t* const w_scale[3],
                                  const float* const bias[3], const float* const gamma[3], void* out16, uint8_t* out8,
                                  uint8_t* out8_scale, float* x32) {
    ME_API_BEGIN(ctx)
    ME_CHECK(A8 && a_scale && W8 && w_scale && bias, ME_ERR_BAD_ARG, "me_op_linear_fp8_segments: null pointer");
    GemmParams p = GemmParams();
    p.M = M, p.N = N, p.K = K, p.A = A8, p.lda = K, p.a_scale = a_scale, p.a_mt = (int)cdiv(M, 128);
    p.W = W8[0], p.w_scale = w_scale[0], p.bias = bias[0], p.ldc = N;
    p.seg1 = seg1, p.seg2 = seg2;
    p.W_s1 = W8[1], p.w_scale_s1 = w_scale[1], p.bias_s1 = bias[1];
    p.W_s2 = W8[2], p.w_scale_s2 = w_scale[2], p.bias_s2 = bias[2];
    p.clamp_lo = -INFINITY, p.clamp_hi = INFINITY;
    if (x32) {
        ME_CHECK(gamma, ME_ERR_BAD_ARG, "me_op_linear_fp8_segments: the residual form takes gamma");
        p.gamma = gamma[0], p.gamma_s1 = gamma[1], p.gamma_s2 = gamma[2], p.res32 = x32, p.out32 = x32;
        gemm_fp8_launch(p, EPI_RESID_SCALE, ctx->stream);
    } else if (out8) {
        p.act = ACT_GELU, p.out8 = out8, p.out8_scale = out8_scale, p.out8_mt = (int)cdiv(M, 128);
        gemm_fp8_launch(p, EPI_STORE, ctx->stream);
    } else {
        p.out16 = out16;
        gemm_fp8_launch(p, EPI_STORE, ctx->stream);
    }
    ME_API_END(ctx)
}

int32_t me_op_linear_fp8_residual_layernorm(me_ctx* ctx, int32_t M, int32_t N, int32_t K, const uint8_t* A8, const uint8_t* a_scale,
                                            int32_t seg1, int32_t seg2, const uint8_t* const W8[3], const uint8_t* const w_scale[3],
                                            const float* const bias[3], const float* const gamma[3], const float* const ln_w[3],
                                            const float* const ln_b[3], float eps, float* x32, uint8_t* xn8, uint8_t* xn_scale) {
    ME_API_BEGIN(ctx)
    ME_CHECK(A8 && a_scale && W8 && w_scale && bias && gamma && ln_w && ln_b && x32 && xn8 && xn_scale, ME_ERR_BAD_ARG,
             "me_op_linear_fp8_residual_layernorm: null pointer");
    const int nseg = seg1 == 0 ? 1 : (seg2 == 0 ? 2 : 3);
    for (int i = 0; i < nseg; ++i)
        ME_CHECK(W8[i] && w_scale[i] && bias[i] && gamma[i] && ln_w[i] && ln_b[i], ME_ERR_BAD_ARG,
                 "me_op_linear_fp8_residual_layernorm: row segment %d without weights", i);
    GemmParams p = GemmParams();
    p.M = M, p.N = N, p.K = K, p.A = A8, p.lda = K, p.a_scale = a_scale, p.a_mt = (int)cdiv(M, 128);
    p.W = W8[0], p.w_scale = w_scale[0], p.bias = bias[0], p.ldc = N;
    p.seg1 = seg1, p.seg2 = seg2;
    p.W_s1 = W8[1], p.w_scale_s1 = w_scale[1], p.bias_s1 = bias[1];
    p.W_s2 = W8[2], p.w_scale_s2 = w_scale[2], p.bias_s2 = bias[2];
    p.clamp_lo = -INFINITY, p.clamp_hi = INFINITY;
    p.gamma = gamma[0], p.gamma_s1 = gamma[1], p.gamma_s2 = gamma[2], p.res32 = x32, p.out32 = x32;
    p.ln_out16 = xn8, p.ln_eps = eps, p.out8 = xn8, p.out8_scale = xn_scale, p.out8_mt = (int32_t)cdiv(M, 128);
    p.ln_w = ln_w[0], p.ln_b = ln_b[0], p.ln_w_s1 = ln_w[1], p.ln_b_s1 = ln_b[1], p.ln_w_s2 = ln_w[2], p.ln_b_s2 = ln_b[2];
    const size_t row_tiles = (size_t)seg_row_tiles<352>(M, seg1, seg2);
    p.ln_stats = (unsigned long long*)site_buf(ctx, "op.ln.stats", row_tiles * (size_t)(N / 256) * 352 * 8);
    p.ln_count = (unsigned*)site_buf(ctx, "op.ln.count." + std::to_string(N), row_tiles * 64);
    gemm_fp8_launch(p, EPI_RESID_SCALE, ctx->stream);
    ME_API_END(ctx)
}

int32_t me_op_linear_segments(me_ctx* ctx, int32_t M, int32_t N, int32_t K, const void* A16, int32_t seg1, int32_t seg2,
                              const void* const W16[3], const float* const bias[3], const float* const gamma[3],
                              void* out16, float* x32, int32_t act, int32_t tile_cfg) {
    ME_API_BEGIN(ctx)
    ME_CHECK(A16 && W16 && bias && (out16 || x32), ME_ERR_BAD_ARG, "me_op_linear_segments: null pointer");
    ME_CHECK(W16[0] && (seg1 == 0 || W16[1]) && (seg2 == 0 || W16[2]), ME_ERR_BAD_ARG,
             "me_op_linear_segments: a row segment without weights");
    ME_CHECK(seg1 >= 0 && seg2 >= 0 && seg1 <= M && seg2 <= M && (seg2 == 0 || (seg1 > 0 && seg2 > seg1)), ME_ERR_BAD_ARG,
             "me_op_linear_segments: segments %d / %d of %d rows", seg1, seg2, M);
    GemmParams p = GemmParams();
    p.M = M, p.N = N, p.K = K, p.A = A16, p.lda = K, p.W = W16[0], p.bias = bias[0], p.ldc = N;
    p.seg1 = seg1, p.seg2 = seg2, p.W_s1 = W16[1], p.bias_s1 = bias[1], p.W_s2 = W16[2], p.bias_s2 = bias[2];
    p.clamp_lo = -INFINITY, p.clamp_hi = INFINITY;
    if (x32) {
        ME_CHECK(gamma && gamma[0] && (seg1 == 0 || gamma[1]) && (seg2 == 0 || gamma[2]), ME_ERR_BAD_ARG,
                 "me_op_linear_segments: the residual form takes gamma for every segment");
        p.gamma = gamma[0], p.gamma_s1 = gamma[1], p.gamma_s2 = gamma[2], p.res32 = x32, p.out32 = x32;
        gemm_launch(p, A_PLAIN, EPI_RESID_SCALE, ctx->dtype, ctx->stream, tile_cfg);
    } else {
        p.out16 = out16, p.act = act;
        gemm_launch(p, A_PLAIN, EPI_STORE, ctx->dtype, ctx->stream, tile_cfg);
    }
    ME_API_END(ctx)
}

namespace {
// xn8 != nullptr: the normalised rows as MX fp8 + activation-layout scales (ceil(M / 128) tiles) instead of 16-bit
void linear_residual_layernorm_impl(me_ctx* ctx, const char* who, int32_t M, int32_t N, int32_t K, const void* A16, int32_t seg1,
                                    int32_t seg2, const void* const W16[3], const float* const bias[3], const float* const gamma[3],
                                    const float* const ln_w[3], const float* const ln_b[3], float eps, float* x32, void* xn16,
                                    uint8_t* xn8, uint8_t* xn_scale) {
    ME_CHECK(A16 && W16 && bias && gamma && ln_w && ln_b && x32 && (xn16 || (xn8 && xn_scale)), ME_ERR_BAD_ARG, "%s: null pointer", who);
    ME_CHECK(seg1 >= 0 && seg2 >= 0 && seg1 <= M && seg2 <= M && (seg2 == 0 || (seg1 > 0 && seg2 > seg1)), ME_ERR_BAD_ARG,
             "%s: segments %d / %d of %d rows", who, seg1, seg2, M);
    const int nseg = seg1 == 0 ? 1 : (seg2 == 0 ? 2 : 3);
    for (int i = 0; i < nseg; ++i)
        ME_CHECK(W16[i] && bias[i] && gamma[i] && ln_w[i] && ln_b[i], ME_ERR_BAD_ARG, "%s: row segment %d without weights", who, i);
    ME_CHECK(N == 256 || N == 512 || N == 1024, ME_ERR_BAD_SHAPE, "%s: N = %d not in {256, 512, 1024}", who, N);
    GemmParams p = GemmParams();
    p.M = M, p.N = N, p.K = K, p.A = A16, p.lda = K, p.W = W16[0], p.bias = bias[0], p.ldc = N;
    p.seg1 = seg1, p.seg2 = seg2, p.W_s1 = W16[1], p.bias_s1 = bias[1], p.W_s2 = W16[2], p.bias_s2 = bias[2];
    p.clamp_lo = -INFINITY, p.clamp_hi = INFINITY;
    p.gamma = gamma[0], p.gamma_s1 = gamma[1], p.gamma_s2 = gamma[2], p.res32 = x32, p.out32 = x32;
    p.ln_out16 = xn8 ? (void*)xn8 : xn16, p.ln_eps = eps;
    if (xn8) p.out8 = xn8, p.out8_scale = xn_scale, p.out8_mt = (int32_t)cdiv(M, 128);
    p.ln_w = ln_w[0], p.ln_b = ln_b[0], p.ln_w_s1 = ln_w[1], p.ln_b_s1 = ln_b[1], p.ln_w_s2 = ln_w[2], p.ln_b_s2 = ln_b[2];
    const size_t row_tiles = (size_t)seg_row_tiles<352>(M, seg1, seg2);
    // (an arrival counter advances by N / 256 per launch and must start a launch at a multiple of that: one per N)
    p.ln_stats = (unsigned long long*)site_buf(ctx, "op.ln.stats", row_tiles * (size_t)(N / 256) * 352 * 8);
    p.ln_count = (unsigned*)site_buf(ctx, "op.ln.count." + std::to_string(N), row_tiles * 64);
    gemm_launch(p, A_PLAIN, EPI_RESID_SCALE, ctx->dtype, ctx->stream, 10);
}
}  // namespace

int32_t me_op_linear_residual_layernorm(me_ctx* ctx, int32_t M, int32_t N, int32_t K, const void* A16, int32_t seg1,
                                        int32_t seg2, const void* const W16[3], const float* const bias[3],
                                        const float* const gamma[3], const float* const ln_w[3], const float* const ln_b[3],
                                        float eps, float* x32, void* xn16) {
    ME_API_BEGIN(ctx)
    ME_CHECK(xn16, ME_ERR_BAD_ARG, "me_op_linear_residual_layernorm: null pointer");
    linear_residual_layernorm_impl(ctx, "me_op_linear_residual_layernorm", M, N, K, A16, seg1, seg2, W16, bias, gamma, ln_w, ln_b, eps,
                                   x32, xn16, nullptr, nullptr);
    ME_API_END(ctx)
}

int32_t me_op_linear_residual_layernorm_fp8(me_ctx* ctx, int32_t M, int32_t N, int32_t K, const void* A16, int32_t seg1,
                                            int32_t seg2, const void* const W16[3], const float* const bias[3],
                                            const float* const gamma[3], const float* const ln_w[3], const float* const ln_b[3],
                                            float eps, float* x32, uint8_t* xn8, uint8_t* xn_scale) {
    ME_API_BEGIN(ctx)
    ME_CHECK(xn8 && xn_scale, ME_ERR_BAD_ARG, "me_op_linear_residual_layernorm_fp8: null pointer");
    linear_residual_layernorm_impl(ctx, "me_op_linear_residual_layernorm_fp8", M, N, K, A16, seg1, seg2, W16, bias, gamma, ln_w, ln_b,
                                   eps, x32, nullptr, xn8, xn_scale);
    ME_API_END(ctx)
}

int32_t me_op_format_f64(me_ctx* ctx, const double* values, int64_t count, char* text, int32_t stride, int32_t* lengths) {
    ME_API_BEGIN(ctx)
    ME_CHECK(values && text && lengths && count >= 0, ME_ERR_BAD_ARG, "me_op_format_f64: bad argument");
    format_f64_launch(values, count, text, stride, lengths, ctx->stream);
    ME_API_END(ctx)
}

int32_t me_op_cast_to16(me_ctx* ctx, const float* src, void* dst16, int64_t count) {
    ME_API_BEGIN(ctx)
    cast_f32_to_16_launch(src, dst16, count, ctx->dtype, ctx->stream);
    ME_API_END(ctx)
}

int32_t me_op_cast_to32(me_ctx* ctx, const void* src16, float* dst, int64_t count) {
    ME_API_BEGIN(ctx)
    cast_16_to_f32_launch(src16, dst, count, ctx->dtype, ctx->stream);
    ME_API_END(ctx)
}

int32_t me_profile_enable(me_ctx* ctx, int32_t on) {
    ME_API_BEGIN(ctx)
    ME_HIP(hipStreamSynchronize(ctx->stream));
    Profiler& p = profiler();
    for (ProfEntry& e : p.entries) {
        (void)hipEventDestroy(e.e0);
        (void)hipEventDestroy(e.e1);
    }
    p.entries.clear();
    p.enabled = on != 0;
    ME_API_END(ctx)
}

int32_t me_profile_report(me_ctx* ctx, char* json, int64_t capacity) {
    ME_API_BEGIN(ctx)
    ME_CHECK(json && capacity > 2, ME_ERR_BAD_ARG, "me_profile_report: no buffer");
    ME_HIP(hipStreamSynchronize(ctx->stream));
    struct Agg {
        double ms = 0, flops = 0, bytes = 0;
        long count = 0;
    };
    std::map<std::string, Agg> agg;
    for (ProfEntry& e : profiler().entries) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.e0, e.e1) != hipSuccess) continue;
        Agg& a = agg[e.name];
        a.ms += ms, a.flops += e.flops, a.bytes += e.bytes, a.count += 1;
    }
    std::string out = "[";
    bool first = true;
    for (auto& kv : agg) {
        char line[512];
        snprintf(line, sizeof line,
                 "%s{\"kernel\": \"%s\", \"launches\": %ld, \"total_ms\": %.6f, \"flops\": %.6e, "
                 "\"bytes\": %.6e}",
                 first ? "" : ", ", kv.first.c_str(), kv.second.count, kv.second.ms, kv.second.flops,
                 kv.second.bytes);
        out += line;
        first = false;
    }
    out += "]";
    ME_CHECK((int64_t)out.size() + 1 <= capacity, ME_ERR_BAD_ARG, "me_profile_report: buffer too small");
    memcpy(json, out.c_str(), out.size() + 1);
    ME_API_END(ctx)
}

int32_t me_op_gemm_config_count(void) { return gemm_num_configs(); }
const char* me_op_gemm_config_name(int32_t cfg) { return gemm_config_name(cfg); }

}  // extern "C"
