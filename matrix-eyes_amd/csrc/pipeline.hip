// The Depth Pro forward pass as a sequence of HIP kernel launches on one stream.
// Stage structure follows reference src/depth_pro/mod.rs:251-363; each function cites the
// module forward() it replaces.  Activations live in HBM as:
//   - ViT: f32 residual stream [windows*tokens][C] + 16-bit GEMM operands
//   - conv stages: NHWC; 16-bit operands of 3x3 convs carry a 1-pixel zero border
//     ([B][H+2][W+2][C]) so the implicit-GEMM loader needs no bounds checks, residual paths
//     stay f32 ([B*H*W][C]).
#include <algorithm>
#include <memory>

#include "model.h"

namespace me {

void report(me_ctx* ctx, float pos, const char* msg) {
    if (ctx->progress)
        ctx->progress(ctx->progress_user, ctx->prog_lo + pos * (ctx->prog_hi - ctx->prog_lo), msg);
}

bool is_device_ptr(const void* p) {
    hipPointerAttribute_t a;
    const hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // plain malloc'd host memory is "invalid value": not an error
        return false;
    }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

void* site_buf(me_ctx* ctx, const std::string& name, size_t bytes) {
    DevBuf& b = ctx->bufs[name];
    if (b.p && b.bytes >= bytes) return b.p;
    if (ctx->capturing) throw CaptureAbort();  // an allocation would synchronise: the caller runs eagerly instead
    if (b.p) {
        ME_HIP(hipStreamSynchronize(ctx->stream));
        ctx->drop_graph();  // a captured step may hold the old address (dropped once its replays have drained)
        ME_HIP(hipFree(b.p));
        b.p = nullptr, b.bytes = 0;
    }
    ME_HIP(hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    ME_HIP(hipMemsetAsync(b.p, 0, bytes, ctx->stream));
    ME_HIP(hipStreamSynchronize(ctx->stream));
    return b.p;
}

const void* to_device(me_ctx* ctx, const void* p, size_t bytes, const std::string& name) {
    if (is_device_ptr(p)) return p;
    void* d = site_buf(ctx, name, bytes);
    ME_HIP(hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, ctx->stream));
    return d;
}

void from_device(me_ctx* ctx, void* dst, const void* src_dev, size_t bytes) {
    if (is_device_ptr(dst)) {
        if (dst != src_dev)
            ME_HIP(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    } else {
        ME_HIP(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
        ME_HIP(hipStreamSynchronize(ctx->stream));
    }
}

namespace {

// ---- thin launch helpers ---------------------------------------------------------------
GemmParams base_params() {
    GemmParams p = GemmParams();  // value-initialised: every pointer null, every int 0
    p.clamp_lo = -INFINITY, p.clamp_hi = INFINITY;
    return p;
}

// Split operands (model.h SplitStage): a_split: A holds [hi | lo] rows of 2K and W is stored [W | W];
// out_split: the 16-bit output is written as [hi | lo] rows of 2N.
void set_out16_split(GemmParams& p, int64_t C) { p.ldc16 = 2 * C, p.lo_off16 = (int32_t)C; }

// Deterministic split-K for the small deep launches of the tail (gemm_core.h gemm_kernel<..., SPLITK>; VERDICT r4
// "missing" 6): a launch whose image has fewer than 200 tiles of 128 x 128 and K >= ME_SPLIT_K_MINK (4096) runs on that tile with
// S work items per tile, S the largest power of two that keeps the launch within 640 work items and a K range of at least 512.
// Decided from ONE image's rows, so a batch takes the decision of its images (a batch stays a loop of batch-one calls bit for
// bit).  OFF unless ME_SPLIT_K=1: measured, the two K = 9216 convolutions it applies to gain 0.02 ms per step together, and with
// a lower K bound (the 96 x 96 level's 3x3 convolutions, the upsample chains' 1x1) the partials' round trip through memory costs
// more than the longer K loop it replaces (+0.15 ms at K >= 2048).
// Returns the tile configuration to force (1) or -1.
int maybe_split_k(me_ctx* ctx, GemmParams& p, int64_t rows_per_image, hipStream_t s) {
    static const bool on = getenv("ME_SPLIT_K") && atoi(getenv("ME_SPLIT_K")) != 0;
    static const int min_k = getenv("ME_SPLIT_K_MINK") ? atoi(getenv("ME_SPLIT_K_MINK")) : 4096;  // (the tests lower it: a tiny model has no K that deep)
    if (!on || rows_per_image <= 0 || p.K < min_k) return -1;
    const int64_t tiles_img = cdiv(rows_per_image, 128) * cdiv((int64_t)p.N, 128);
    if (tiles_img >= 200) return -1;
    int S = 1;
    while (S < 8 && tiles_img * (S * 2) <= 640 && p.K / (S * 2) >= 512) S *= 2;
    if (S == 1) return -1;
    const int64_t tiles = cdiv((int64_t)p.M, 128) * cdiv((int64_t)p.N, 128);
    if (tiles > 4096) return -1;
    // one workspace per stream (launches of one stream follow one another; a second stream must not share it)
    const std::string tag = "splitk." + std::to_string((uintptr_t)s);
    p.split_k = S;
    p.splitk_ws = (float*)site_buf(ctx, tag + ".ws", (size_t)tiles * S * 128 * 128 * 4);   // grows to the largest launch
    p.splitk_cnt = (unsigned*)site_buf(ctx, tag + ".cnt", 4096 * 4);   // zero when allocated; the last arriver of a tile resets its word
    return 1;
}

// out = act(A[M][K] . W[N][K]^T + bias) as 16-bit and/or f32 rows of stride ldc
// rows_per_image > 0: the launch may take the split-K form (maybe_split_k)
void linear(me_ctx* ctx, const void* A, int64_t M, int K, const void* W, int N, const float* bias,
            void* out16, float* out32, int64_t ldc, int act, hipStream_t s, bool a_split = false,
            bool out_split = false, int qcols = 0, int64_t rows_per_image = 0) {
    GemmParams p = base_params();
    const int Kx = a_split ? 2 * K : K;
    p.M = (int)M, p.N = N, p.K = Kx, p.flop_k = K, p.A = A, p.lda = Kx, p.W = W, p.bias = bias;
    p.out16 = out16, p.out32 = out32, p.ldc = ldc, p.act = act;
    p.qcols = qcols, p.qscale = kAttnQScale;  // the qkv linear: Q leaves scaled for the attention kernel
    p.grid_cap = ctx->grid_cap;
    if (out_split) set_out16_split(p, N);
    gemm_launch(p, A_PLAIN, EPI_STORE, ctx->dtype, s, maybe_split_k(ctx, p, rows_per_image, s));
}

struct ConvOut {
    float* out32 = nullptr;   // [B*H*W][Cout]
    void* out16 = nullptr;    // 16-bit copy
    bool border16 = false;    // out16 is [B][H+2][W+2][Cout]
    bool split16 = false;     // out16 pixels are [hi | lo] of 2*Cout
    bool triple16 = false;    // ... [hi | lo | hi] of 3*Cout (with split16)
    int act = ACT_NONE;       // applied to out16 (and out32 unless act16_only)
    bool act16_only = true;
    const float* res32 = nullptr;
    const float* res32b = nullptr;
    const float* tap_bias = nullptr;  // GemmParams::tap_bias ([9][Cout]; needs border16)
};

// Conv2d k x k (k = 1 or 3, pad (k-1)/2, stride 1 or 2) on a zero-bordered NHWC operand
void conv(me_ctx* ctx, const void* in16b, int B, int Hin, int Win, int Cin, const void* W, int Cout,
          int k, int stride, const float* bias, const ConvOut& o, hipStream_t s, bool a_split = false) {
    GemmParams p = base_params();
    const int Ho = Hin / stride, Wo = Win / stride;
    p.flop_k = k * k * Cin;
    if (a_split) Cin *= 2;  // pixels of [hi | lo]; the packed weights repeat each tap's Cin values
    p.M = B * Ho * Wo, p.N = Cout, p.K = k * k * Cin;
    p.A = in16b, p.in_Hp = Hin + 2, p.in_Wp = Win + 2, p.Cin = Cin;
    p.out_H = Ho, p.out_W = Wo, p.KH = k, p.KW = k, p.stride = stride;
    p.W = W, p.bias = bias, p.res32 = o.res32, p.res32b = o.res32b, p.tap_bias = o.tap_bias;
    p.out32 = o.out32, p.out16 = o.out16, p.ldc = Cout, p.out16_border = o.border16 ? 1 : 0;
    p.act = o.act, p.act16_only = o.act16_only ? 1 : 0;
    if (o.split16) set_out16_split(p, Cout);
    if (o.triple16) p.ldc16 = 3 * Cout, p.hi2_off16 = 2 * Cout;
    p.grid_cap = ctx->grid_cap;
    gemm_launch(p, A_CONV, EPI_STORE, ctx->dtype, s, o.tap_bias ? -1 : maybe_split_k(ctx, p, (int64_t)Ho * Wo, s));
}

// ConvTranspose2d(2,2,stride 2) of an unbordered NHWC operand [B*H*W][Cin] -> [B][2H][2W][Cout]
// pixel_stride: channels per pixel of out16 when it is a slice of a wider map (0: Cout, or 2 Cout when split);
// out_split: out16 pixels are [hi | lo], the lo part lo_off channels after the hi part (0: Cout)
void convt(me_ctx* ctx, const void* in16, int B, int H, int W_, int Cin, const void* W, int Cout,
           const float* bias, float* out32, void* out16, bool border16, int64_t pixel_stride,
           int act16, hipStream_t s, bool a_split = false, bool out_split = false, int64_t lo_off = 0,
           int k_copies = 0 /* > 0: A rows hold this many Cin-wide parts ([hi | lo | hi] = 3) */) {
    GemmParams p = base_params();
    const int Kx = k_copies > 0 ? k_copies * Cin : (a_split ? 2 * Cin : Cin);
    p.M = B * H * W_, p.N = 4 * Cout, p.K = Kx, p.A = in16, p.lda = Kx, p.W = W, p.bias = bias;
    // the composed deconv o out_conv (k_copies == 3) stands for two layers of SURVEY App. B: ConvT (Cin x 4 Cout per
    // input pixel) and the 1x1 conv at 4x the pixels (Cout x Cout each) -- twice the ConvT's FLOPs when Cin == Cout
    p.flop_k = k_copies == 3 ? 2 * Cin : Cin;
    p.out_H = H, p.out_W = W_, p.Cout = Cout, p.out32 = out32, p.out16 = out16;
    p.out16_border = border16 ? 1 : 0, p.ldc = Cout, p.act = act16;
    if (out_split) {
        p.lo_off16 = (int32_t)(lo_off ? lo_off : Cout);
        p.ldc16 = pixel_stride ? pixel_stride : 2 * Cout;
    } else if (pixel_stride) {
        p.ldc16 = pixel_stride;
    }
    p.grid_cap = ctx->grid_cap;
    gemm_launch(p, A_PLAIN, EPI_CONVT, ctx->dtype, s);
}

size_t bordered_bytes(int B, int H, int W, int C) { return (size_t)B * (H + 2) * (W + 2) * C * 2; }

// encoder.rs:210-216 forward_seq over an upsample block: 1x1 conv (no bias) then ConvT x n.
// The last ConvT writes the caller's outputs.  With SPLIT_UPSAMPLE the input and every intermediate are
// [hi | lo] operands; last_split / last_lo_off say whether (and where) the last 16-bit output is split too.
void run_upsample(me_ctx* ctx, const std::string& tag, const void* in16, int B, int H,
                  const UpsampleW& u, float* last32, void* last16, bool last_border,
                  int64_t last_pixel_stride, int last_act16, hipStream_t s, bool last_split = false,
                  int64_t last_lo_off = 0) {
    const int C = ctx->C();
    const bool sp = ctx->split(SPLIT_UPSAMPLE);
    const int64_t M = (int64_t)B * H * H;
    const size_t wide = sp ? 2 : 1;
    void* a = site_buf(ctx, tag + ".proj", (size_t)M * u.dim_int * 2 * wide);
    linear(ctx, in16, M, C, u.conv, u.dim_int, nullptr, a, nullptr, u.dim_int, ACT_NONE, s, sp, sp, 0, (int64_t)H * H);
    const void* cur = a;
    int h = H;
    const int n = (int)u.convt.size();
    for (int i = 0; i < n; ++i) {
        const bool last = i == n - 1;
        if (last) {
            convt(ctx, cur, B, h, h, u.cin[i], u.convt[i], u.cout[i], nullptr, last32, last16,
                  last_border, last_pixel_stride, last_act16, s, sp, last_split, last_lo_off);
        } else {
            void* t = site_buf(ctx, tag + ".up" + std::to_string(i),
                               (size_t)B * 4 * h * h * u.cout[i] * 2 * wide);
            convt(ctx, cur, B, h, h, u.cin[i], u.convt[i], u.cout[i], nullptr, nullptr, t, false, 0,
                  ACT_NONE, s, sp, sp);
            cur = t;
        }
        h *= 2;
    }
}

struct TapCtx {
    me_ctx* ctx;
    int B;
    void *lat0, *lat1;
    hipStream_t s;
};

void tap_fn(void* user, int index, const float* tokens) {
    TapCtx* t = (TapCtx*)user;
    me_ctx* ctx = t->ctx;
    // encoder.rs:266-280: reshape_feature + merge of the first 25 windows, padding 3 (= g/8)
    void* dst = index == ctx->cfg.tap_blocks[0] ? t->lat0
                                                : (index == ctx->cfg.tap_blocks[1] ? t->lat1 : nullptr);
    if (dst)
        merge_launch(tokens, nullptr, dst, t->B, 35, 0, 5, ctx->g() / 8, ctx->g(), ctx->C(),
                     ctx->dtype, t->s, ctx->split(SPLIT_UPSAMPLE) ? 1 : 0);
}

}  // namespace

// vit.rs:287-346: prepare_tokens_with_mask, 24 x Block::forward, final LayerNorm -- as an object, so
// that the encoder can issue the three ViTs block by block in turn (see stage_encoder).
namespace {
struct VitRun {
    me_ctx* ctx;
    const VitW& v;
    int W;
    VitTaps taps;
    hipStream_t s;
    int64_t rows;
    float* tok;
    void *xn, *qkv, *att, *hid;

    // vit.rs:287-295: patch embed + cls + pos
    VitRun(me_ctx* ctx_, int which, const void* patches16, int W_, const VitTaps& taps_, const std::string& tag,
           hipStream_t s_)
        : ctx(ctx_), v(ctx_->w.vit[which]), W(W_), taps(taps_), s(s_) {
        const int C = ctx->C(), T = ctx->T(), P = ctx->P();
        rows = (int64_t)W * T;
        tok = (float*)site_buf(ctx, tag + ".tokens", (size_t)rows * C * 4);
        xn = site_buf(ctx, tag + ".xn", (size_t)rows * C * 2);
        qkv = site_buf(ctx, tag + ".qkv", (size_t)rows * 3 * C * 2);
        att = site_buf(ctx, tag + ".att", (size_t)rows * C * 2);
        hid = site_buf(ctx, tag + ".hid", (size_t)rows * 4 * C * 2);
        cls_rows_launch(tok, v.cls, v.pos, W, T, C, s);
        GemmParams p = base_params();
        p.M = W * P, p.N = C, p.K = 768, p.A = patches16, p.lda = 768, p.W = v.patch_w;
        p.bias = v.patch_b, p.pos = v.pos, p.out32 = tok, p.ldc = C, p.tokens_per_window = P;
        gemm_launch(p, A_PLAIN, EPI_PATCH_EMBED, ctx->dtype, s);
    }

    // vit.rs:163-170 Block::forward
    void block(int i) {
        const VitBlockW& b = v.blocks[i];
        const int C = ctx->C(), T = ctx->T(), heads = ctx->cfg.num_heads;
        layernorm_launch(tok, b.ln1_w, b.ln1_b, xn, nullptr, rows, C, ctx->cfg.ln_eps, ctx->dtype, s);
        linear(ctx, xn, rows, C, b.qkv_w, 3 * C, b.qkv_b, qkv, nullptr, 3 * C, ACT_NONE, s, false, false, C);
        attention_launch(qkv, att, W, T, heads, ctx->dtype, s, nullptr, nullptr, nullptr, 0, true);
        {
            GemmParams p = base_params();
            p.M = (int)rows, p.N = C, p.K = C, p.A = att, p.lda = C, p.W = b.proj_w;
            p.bias = b.proj_b, p.gamma = b.ls1, p.res32 = tok, p.out32 = tok, p.ldc = C;
            gemm_launch(p, A_PLAIN, EPI_RESID_SCALE, ctx->dtype, s);
        }
        layernorm_launch(tok, b.ln2_w, b.ln2_b, xn, nullptr, rows, C, ctx->cfg.ln_eps, ctx->dtype, s);
        linear(ctx, xn, rows, C, b.fc1_w, 4 * C, b.fc1_b, hid, nullptr, 4 * C, ACT_GELU, s);
        {
            GemmParams p = base_params();
            p.M = (int)rows, p.N = C, p.K = 4 * C, p.A = hid, p.lda = 4 * C, p.W = b.fc2_w;
            p.bias = b.fc2_b, p.gamma = b.ls2, p.res32 = tok, p.out32 = tok, p.ldc = C;
            gemm_launch(p, A_PLAIN, EPI_RESID_SCALE, ctx->dtype, s);
        }
        if (taps.fn) taps.fn(taps.user, i, tok);
    }

    // vit.rs:343 final norm
    void finish(void* final16, float* final32) {
        layernorm_launch(tok, v.norm_w, v.norm_b, final16, final32, rows, ctx->C(), ctx->cfg.ln_eps, ctx->dtype, s);
    }
};

// The encoder's three ViTs (patch encoder on 35 B windows, image encoder and FOV encoder on B windows each,
// different weights, same architecture) as ONE row space: segment 0 = the patch encoder's rows, 1 = the
// image encoder's, 2 = the FOV encoder's, each padded to a multiple of 256 rows so that no GEMM tile
// straddles two weight sets.  Every kernel of a block then runs ONCE over all rows, with the weights chosen
// per row tile (GEMMs, GemmParams::seg1/seg2), per row (LayerNorm) or not at all (attention: a window ->
// row map).  qkv: 948 + 72 tiles still fit the four rounds the patch encoder needs alone; fc1 goes from
// 4.94 to 5.31 rounds, which costs what the separate small launches did; proj / fc2 move to the 256x256
// tile (see resid_all).  This replaces two side streams of ~170 small launches each, which took CUs from
// the launch stream's persistent kernels at every kernel boundary (fc2 0.195 -> 0.34 ms on most layers):
// 26.5 -> 25.5 ms per step, 580 -> 390 launches.
struct MergedVit {
    me_ctx* ctx;
    const VitW &v0, &v1, &v2;  // patch encoder, image encoder, FOV encoder
    // Physical order of the row space: the SMALL segments first -- [image encoder | FOV encoder | patch encoder]
    // (without the FOV head: [image encoder | patch encoder]).  Every boundary is then a multiple of the small
    // segments' padded length (768 rows at one image: a multiple of 256 AND of 192, so the 192-row residual tile
    // can be used), and the one ragged end is the end of the row space, which every kernel clamps anyway.
    const VitW &p0, &p1, &p2;  // the ViT of physical segment 0, 1, 2 (p2 unused without the FOV head)
    int W0, W1;        // windows of the patch encoder and of each small ViT
    bool fov;
    VitTaps taps;
    hipStream_t s;
    int64_t R0, seg1, seg2, Rtot;  // rows of the patch encoder; starts of physical segments 1 and 2 (seg2 == 0: two)
    int64_t row_main, row_img, row_fov;  // first row of each ViT
    float* tok;
    char *xn, *qkv, *att, *hid, *fin16;
    bool xn_is_norm1 = false;  // xn already holds norm1 of the block about to run (written by the block before)
    bool xn_is_norm1_fp8 = false;  // fp8 contexts: xn / xn_s hold norm1 of the block about to run as MX fp8 (the fp8 fc2's epilogue)
    bool xn_is_norm2_fp8 = false;  // fp8 contexts: xn / xn_s hold norm2 as MX fp8 (written by the projection's epilogue)
    float* fin32 = nullptr;
    // ME_DTYPE_FP8: xn and hid hold e4m3 bytes, with their block scales (activation layout, Rtot / 128 tiles)
    uint8_t *xn_s = nullptr, *hid_s = nullptr, *att8 = nullptr, *att_s = nullptr;
    RowSegs segs;

    static int64_t pad256(int64_t r) { return (r + 255) / 256 * 256; }
    // rows that hold tokens (the profiler's algorithmic FLOPs; the padding rows of each segment are not work)
    int64_t real_rows() const { return (int64_t)(W0 + W1 * (fov ? 2 : 1)) * ctx->T(); }

    MergedVit(me_ctx* ctx_, int B, bool fov_, const void* patches_main, const void* patches_img,
              const void* patches_fov, const VitTaps& taps_, hipStream_t s_)
        : ctx(ctx_), v0(ctx_->w.vit[ME_VIT_PATCH_ENCODER]), v1(ctx_->w.vit[ME_VIT_IMAGE_ENCODER]),
          v2(ctx_->w.vit[ME_VIT_FOV_ENCODER]), p0(v1), p1(fov_ ? v2 : v0), p2(v0), W0(35 * B), W1(B), fov(fov_),
          taps(taps_), s(s_) {
        const int C = ctx->C(), T = ctx->T(), P = ctx->P();
        R0 = (int64_t)W0 * T;
        const int64_t side = pad256((int64_t)W1 * T);
        seg1 = side;
        seg2 = fov ? 2 * side : 0;
        row_img = 0, row_fov = fov ? side : -1, row_main = fov ? 2 * side : side;
        Rtot = row_main + pad256(R0);
        tok = (float*)site_buf(ctx, "vitm.tokens", (size_t)Rtot * C * 4);
        xn = (char*)site_buf(ctx, "vitm.xn", (size_t)Rtot * C * 2);
        qkv = (char*)site_buf(ctx, "vitm.qkv", (size_t)Rtot * 3 * C * 2);
        att = (char*)site_buf(ctx, "vitm.att", (size_t)Rtot * C * 2);
        hid = (char*)site_buf(ctx, "vitm.hid", (size_t)Rtot * 4 * C * 2);
        if (ctx->fp8) {
            xn_s = (uint8_t*)site_buf(ctx, "vitm.xn.scale", (size_t)Rtot * C / 32);
            hid_s = (uint8_t*)site_buf(ctx, "vitm.hid.scale", (size_t)Rtot * 4 * C / 32);
            att8 = (uint8_t*)site_buf(ctx, "vitm.att8", (size_t)Rtot * C);
            att_s = (uint8_t*)site_buf(ctx, "vitm.att.scale", (size_t)Rtot * C / 32);
        }
        fin16 = (char*)site_buf(ctx, "vitm.final16", (size_t)Rtot * C * 2);
        // attention: windows of physical segments 0 and 1, the rest belong to the last segment
        segs.seg1 = seg1, segs.seg2 = seg2, segs.win0 = W1, segs.win1 = fov ? W1 : W0;
        // vit.rs:287-295 per ViT: patch embed + cls + pos into its segment
        embed(v0, patches_main, W0, row_main);
        embed(v1, patches_img, W1, row_img);
        if (fov) embed(v2, patches_fov, W1, row_fov);
        // The padding rows of each segment take part in every row-wise kernel of a block (x += gamma * (...),
        // 48 times per call) and the buffer outlives the call: zeroed here, they hold the same finite values on
        // every call instead of growing without bound or keeping another batch size's rows.
        zero_rows(row_img + (int64_t)W1 * T, row_img + side);
        if (fov) zero_rows(row_fov + (int64_t)W1 * T, row_fov + side);
        zero_rows(row_main + R0, Rtot);
        (void)P;
    }

    void zero_rows(int64_t r0, int64_t r1) {
        if (r1 > r0) ME_HIP(hipMemsetAsync(tok + r0 * ctx->C(), 0, (size_t)(r1 - r0) * ctx->C() * 4, s));
    }

    void embed(const VitW& v, const void* patches16, int W, int64_t row) {
        const int C = ctx->C(), T = ctx->T(), P = ctx->P();
        cls_rows_launch(tok + row * C, v.cls, v.pos, W, T, C, s);
        GemmParams p = base_params();
        p.M = W * P, p.N = C, p.K = 768, p.A = patches16, p.lda = 768, p.W = v.patch_w;
        p.bias = v.patch_b, p.pos = v.pos, p.out32 = tok + row * C, p.ldc = C, p.tokens_per_window = P;
        gemm_launch(p, A_PLAIN, EPI_PATCH_EMBED, ctx->dtype, s);
    }

    void set_ln(const float* w1, const float* b1, const float* w2, const float* b2) {
        segs.w1 = w1, segs.b1 = b1, segs.w2 = w2, segs.b2 = b2;
    }

    // Whole rounds + a short tail.  A persistent 256x256 launch runs ceil(tiles / 256) rounds, and at one image the last
    // round of fc1 (1360 tiles: 5.31 rounds) and of proj / fc2 (340 tiles: 1.33 rounds) is a third full.  The main loops
    // are bound by what a CU can stage into LDS per slab ((BM + BN) x 128 bytes at ~50 GB/s per CU), not by the MFMAs,
    // so a 96x256 tile costs 0.69 of a 256x256 tile's main loop and 0.375 of its epilogue: when the last round is at
    // most half full, its rows run as a SECOND launch of 96-row tiles (tile config 7; <= 256 of them, one short round
    // in which every workgroup has a tile) behind the whole rounds.  The rows of the second launch all belong to the
    // last row segment.  Results are bit-identical (same K order per output element).  Measured at one image, FOV head:
    // proj 93.7 -> 82.7 us, fc2 211.7 -> 207.7 us stand-alone (profiles/r03_tail_tile_ab.txt); in the step, same box,
    // alternating runs: 25.17 ms (off) -> 24.90 (proj / fc2 split) -> 24.70 (fc1 too).  ME_GEMM_TAIL96=0 turns it off.
    bool launch_with_short_tail(const GemmParams& p, EpiKind epi, const void* w1, const float* b1, const float* g1,
                                const void* w2, const float* b2, const float* g2) {
        static const bool enabled = !(getenv("ME_GEMM_TAIL96") && atoi(getenv("ME_GEMM_TAIL96")) == 0);
        if (!enabled || p.N % 256 || p.M % 256 || p.K < 128) return false;
        const int64_t nbn = p.N / 256;
        if (256 % nbn) return false;
        const int64_t per_round = 256 / nbn, row_tiles = p.M / 256;      // row tiles per round of 256 workgroups
        const int64_t rounds = row_tiles / per_round, rem = row_tiles % per_round;
        const int64_t rows1 = rounds * per_round * 256, rows2 = p.M - rows1;
        const int64_t last_seg = p.seg2 ? p.seg2 : p.seg1;
        if (rounds < 1 || rem == 0 || 2 * rem > per_round || cdiv(rows2, 96) * nbn > 256 || rows1 < last_seg) return false;
        GemmParams a = p, b = p;
        a.M = (int)rows1;
        a.flop_rows = (int)std::min<int64_t>(real_rows(), rows1);
        b.M = (int)rows2;
        b.flop_rows = (int)std::max<int64_t>(0, real_rows() - rows1);
        b.A = (const char*)p.A + rows1 * p.lda * 2;
        if (p.out16) b.out16 = (char*)p.out16 + rows1 * p.ldc * 2;
        if (p.out32) b.out32 = p.out32 + rows1 * p.ldc;
        if (p.res32) b.res32 = p.res32 + rows1 * p.ldc;
        b.seg1 = b.seg2 = 0;  // one weight set: the last segment's
        if (p.seg2) b.W = w2, b.bias = b2, b.gamma = g2;
        else if (p.seg1) b.W = w1, b.bias = b1, b.gamma = g1;
        gemm_launch(a, A_PLAIN, epi, ctx->dtype, s, 0);
        gemm_launch(b, A_PLAIN, epi, ctx->dtype, s, 7);
        return true;
    }

    // The 352-row tile (tile config 10).  256 CUs x one tile each cover 64 x 352 = 22528 rows of an N = 1024 launch: at
    // one image proj / fc2 are ONE exact round (3 + 3 + 58 row tiles x 4 = 256 tiles) instead of a round of 256x256
    // tiles and a second launch of short tiles that the CU cannot stage fast enough (1.375 tile-times against 1 + 0.69),
    // and fc1 is four rounds instead of five and a tail.  Chosen when its rounds x 352 undercut what the 256-row path
    // costs (whole rounds x 256, + 0.69 x 256 for a short tail or a whole round where no tail applies).
    // ME_GEMM_TALL=0 turns it off.
    bool tall_tile_wins(const GemmParams& p) const {
        static const bool enabled = !(getenv("ME_GEMM_TALL") && atoi(getenv("ME_GEMM_TALL")) == 0);
        if (!enabled || p.N % 256 || p.K < 128 || ctx->C() < 256) return false;
        const int64_t nbn = p.N / 256;
        const int64_t tiles352 = (int64_t)seg_row_tiles<352>(p.M, p.seg1, p.seg2) * nbn;
        const int64_t tiles256 = cdiv(p.M, 256) * nbn;
        const double cost352 = (double)cdiv(tiles352, 256) * 352.0;
        const int64_t whole = tiles256 / 256, rem = tiles256 % 256;
        const bool short_tail = whole >= 1 && rem != 0 && 2 * rem <= 256 && 256 % nbn == 0;
        const double cost256 = ((double)whole + (rem == 0 ? 0.0 : (short_tail ? 0.69 : 1.0))) * 256.0;
        // a near-tie goes to the tall tile: qkv's 768 tall tiles (three exact rounds, 1056) against 1020 tiles of
        // 256 x 256 (3.98 rounds, 1024) measured 0.5 % faster in the step (23.27 -> 23.16 ms, three alternating runs)
        static const double bias = getenv("ME_GEMM_TALL_BIAS") ? atof(getenv("ME_GEMM_TALL_BIAS")) : 0.95;
        return cost352 * bias < cost256;
    }

    // one GEMM over all segments (three weight sets): 16-bit output (qkv, fc1) ...
    void gemm_all(const void* A, int K, const void* w0, const void* w1, const void* w2, const float* b0,
                  const float* b1, const float* b2, int N, void* out16, int act, int qcols = 0) {
        GemmParams p = base_params();
        p.M = (int)Rtot, p.N = N, p.K = K, p.A = A, p.lda = K, p.W = w0, p.bias = b0;
        p.qcols = qcols, p.qscale = kAttnQScale;  // qkv: Q leaves scaled for the attention kernel
        p.flop_rows = (int)real_rows();
        p.out16 = out16, p.ldc = N, p.act = act;
        p.seg1 = (int)seg1, p.seg2 = (int)seg2, p.W_s1 = w1, p.bias_s1 = b1, p.W_s2 = w2, p.bias_s2 = b2;
        // fc1 at one image: five whole rounds + 224 short tiles instead of a sixth round a third full (qkv's 1020 tiles
        // are 3.98 rounds and N = 3072 does not divide the round: launch_with_short_tail declines)
        if (tall_tile_wins(p)) {
            gemm_launch(p, A_PLAIN, EPI_STORE, ctx->dtype, s, 10);
            return;
        }
        if (ctx->C() >= 256 && launch_with_short_tail(p, EPI_STORE, w1, b1, nullptr, w2, b2, nullptr)) return;
        gemm_launch(p, A_PLAIN, EPI_STORE, ctx->dtype, s);
    }
    // ... or the residual update x += gamma * (A W^T + b) (proj, fc2).  On the two-group 256x256 tile: the
    // patch encoder's rows alone take 160x128 (two exact rounds), but with the small segments' 24 tiles more
    // the 340 tiles of 256x256 (two rounds) win: 25.5 vs 26.0 ms per step against separate launches.
    // LayerNorm weights of the three physical segments for a launch that normalises the rows it updates
    struct LnSet {
        const float *w0 = nullptr, *b0 = nullptr, *w1 = nullptr, *b1 = nullptr, *w2 = nullptr, *b2 = nullptr;
    };
    // The residual update and the LayerNorm behind it in one launch (gemm_core.h resid_ln_epilogue): 16-bit contexts, the
    // 352-row tile, C in {256, 512, 1024}.  When it applies it applies at EVERY batch size (the tall tile is then taken
    // whatever tall_tile_wins says), so that a row's statistics are summed in the same order whatever batch the row is
    // part of: a batch stays a loop of batch-one calls bit for bit.  ME_LN_FUSE=0: the stand-alone LayerNorm launches.
    bool ln_fusable() const {
        static const bool enabled = !(getenv("ME_LN_FUSE") && atoi(getenv("ME_LN_FUSE")) == 0);
        static const bool tall = !(getenv("ME_GEMM_TALL") && atoi(getenv("ME_GEMM_TALL")) == 0);
        static const bool pp192 = getenv("ME_GEMM_PP192") != nullptr;
        const int C = ctx->C();
        if (!(enabled && tall && !pp192 && (C == 256 || C == 512 || C == 1024))) return false;
        // the context has seen the exchange time out (api.hip run_with_ln_fallback): stand-alone LayerNorm from then on
        if (ctx->ln_fuse_off) return false;
        // one row tile per XCD with all its column tiles must be resident at once on the CUs the stream may use
        // (a smaller part, a partition mode, a stream with a CU mask): otherwise the stand-alone LayerNorm
        return gemm_lnf_resident(ctx->dtype, s) >= 8 * (C / 256);
    }
    // returns true when xn holds LayerNorm(ln) of the updated rows
    bool resid_all(const char* A, int K, const void* w0, const float* bb0, const float* g0, const void* w1,
                   const float* bb1, const float* g1, const void* w2, const float* bb2, const float* g2,
                   const LnSet* ln = nullptr, bool ln_fp8 = false /* xn / xn_s as the MX fp8 operand of the next GEMM */) {
        const int C = ctx->C();
        GemmParams p = base_params();
        p.M = (int)Rtot, p.N = C, p.K = K, p.A = A, p.lda = K, p.W = w0, p.bias = bb0, p.gamma = g0;
        p.flop_rows = (int)real_rows();
        p.res32 = tok, p.out32 = tok, p.ldc = C;
        p.seg1 = (int)seg1, p.seg2 = (int)seg2, p.W_s1 = w1, p.bias_s1 = bb1, p.gamma_s1 = g1;
        p.W_s2 = w2, p.bias_s2 = bb2, p.gamma_s2 = g2;
        // the two-group 256x256 kernel (the cost model rates 128x128 a hair cheaper here; measured it is not).  Its
        // 192-row form (tile config 5; the segment order above makes every boundary a multiple of 192 at one image)
        // wins for proj stand-alone (94.7 -> 80.3 us) and loses in the step (90 -> 95.6 us per launch, same box,
        // bench.py A/B): ME_GEMM_PP192=1 selects it for that comparison.
        static const bool use_pp192 = getenv("ME_GEMM_PP192") != nullptr;
        const bool pp = C >= 256 && K >= 128;
        const bool pp192 = pp && K <= 1024 && use_pp192 && seg1 % 192 == 0 && seg2 % 192 == 0;
        if (ln && pp && ln_fusable() && (ln_fp8 || !ctx->fp8)) {
            const size_t row_tiles = (size_t)seg_row_tiles<352>((int)Rtot, (int)seg1, (int)seg2);
            p.ln_out16 = xn, p.ln_eps = ctx->cfg.ln_eps;
            if (ln_fp8) p.out8 = (uint8_t*)xn, p.out8_scale = xn_s, p.out8_mt = (int)(Rtot / 128);
            p.ln_w = ln->w0, p.ln_b = ln->b0, p.ln_w_s1 = ln->w1, p.ln_b_s1 = ln->b1, p.ln_w_s2 = ln->w2, p.ln_b_s2 = ln->b2;
            p.ln_stats = (unsigned long long*)site_buf(ctx, "vitm.ln.stats", row_tiles * (size_t)(C / 256) * 352 * 8);
            p.ln_count = (unsigned*)site_buf(ctx, "vitm.ln.count", row_tiles * 64);   // zero when allocated, never reset
            gemm_launch(p, A_PLAIN, EPI_RESID_SCALE, ctx->dtype, s, 10);
            return true;
        }
        if (pp && !pp192 && tall_tile_wins(p)) {
            gemm_launch(p, A_PLAIN, EPI_RESID_SCALE, ctx->dtype, s, 10);
            return false;
        }
        if (pp && !pp192 && launch_with_short_tail(p, EPI_RESID_SCALE, w1, bb1, g1, w2, bb2, g2)) return false;
        gemm_launch(p, A_PLAIN, EPI_RESID_SCALE, ctx->dtype, s, pp192 ? 5 : (pp ? 0 : -1));
        return false;
    }

    // the MX fp8 form of gemm_all / resid_all (gemm_fp8.hip): A = xn or hid as e4m3 + block scales
    // ln (fc2 / proj only): also write LayerNorm(ln) of the updated rows into xn / xn_s as MX fp8 -- the tall tile's residual
    // epilogue; returns true when it did
    bool gemm8(const void* A8, const uint8_t* As, int K, int N, const VitBlockW& b0, const VitBlockW& b1,
               const VitBlockW& b2, int which /*0 qkv, 1 fc1, 2 fc2, 3 proj*/, bool hid16_out = false, const LnSet* ln = nullptr) {
        GemmParams p = base_params();
        auto w8 = [&](const VitBlockW& b) { return which == 0 ? b.qkv_w8 : (which == 1 ? b.fc1_w8 : (which == 2 ? b.fc2_w8 : b.proj_w8)); };
        auto ws = [&](const VitBlockW& b) { return which == 0 ? b.qkv_ws : (which == 1 ? b.fc1_ws : (which == 2 ? b.fc2_ws : b.proj_ws)); };
        auto bias = [&](const VitBlockW& b) { return which == 0 ? b.qkv_b : (which == 1 ? b.fc1_b : (which == 2 ? b.fc2_b : b.proj_b)); };
        auto gamma = [&](const VitBlockW& b) { return which == 3 ? b.ls1 : b.ls2; };
        p.M = (int)Rtot, p.N = N, p.K = K, p.A = A8, p.lda = K, p.a_scale = As, p.a_mt = (int)(Rtot / 128);
        p.flop_rows = (int)real_rows();
        p.W = w8(b0), p.w_scale = ws(b0), p.bias = bias(b0), p.ldc = N;
        p.seg1 = (int)seg1, p.seg2 = (int)seg2;
        p.W_s1 = w8(b1), p.w_scale_s1 = ws(b1), p.bias_s1 = bias(b1);
        p.W_s2 = w8(b2), p.w_scale_s2 = ws(b2), p.bias_s2 = bias(b2);
        if (which == 0) {
            p.out16 = qkv;
            p.qcols = ctx->C(), p.qscale = kAttnQScale;
            gemm_fp8_launch(p, EPI_STORE, s);
        } else if (which == 1) {
            p.act = ACT_GELU;
            if (hid16_out) p.out16 = hid;  // fc2 stays 16-bit: the hidden layer leaves as f16
            else p.out8 = (uint8_t*)hid, p.out8_scale = hid_s, p.out8_mt = (int)(Rtot / 128);
            gemm_fp8_launch(p, EPI_STORE, s);
        } else {
            p.gamma = gamma(b0), p.gamma_s1 = gamma(b1), p.gamma_s2 = gamma(b2), p.res32 = tok, p.out32 = tok;
            static const bool fp8_lnf = !(getenv("ME_FP8_LN_FUSE") && atoi(getenv("ME_FP8_LN_FUSE")) == 0);
            if (ln && fp8_lnf && ln_fusable()) {
                const int C = ctx->C();
                const size_t row_tiles = (size_t)seg_row_tiles<352>((int)Rtot, (int)seg1, (int)seg2);
                p.ln_out16 = xn, p.ln_eps = ctx->cfg.ln_eps;
                p.out8 = (uint8_t*)xn, p.out8_scale = xn_s, p.out8_mt = (int)(Rtot / 128);
                p.ln_w = ln->w0, p.ln_b = ln->b0, p.ln_w_s1 = ln->w1, p.ln_b_s1 = ln->b1, p.ln_w_s2 = ln->w2, p.ln_b_s2 = ln->b2;
                p.ln_stats = (unsigned long long*)site_buf(ctx, "vitm.ln.stats", row_tiles * (size_t)(C / 256) * 352 * 8);
                p.ln_count = (unsigned*)site_buf(ctx, "vitm.ln.count", row_tiles * 64);   // zero when allocated, never reset
                gemm_fp8_launch(p, EPI_RESID_SCALE, s);
                return true;
            }
            gemm_fp8_launch(p, EPI_RESID_SCALE, s);
        }
        return false;
    }

    // vit.rs:163-170 Block::forward for the three ViTs
    void block(int i) {
        const VitBlockW &b0 = p0.blocks[i], &b1 = p1.blocks[i], &b2 = p2.blocks[i];  // by physical segment
        const int C = ctx->C(), T = ctx->T(), heads = ctx->cfg.num_heads;
        if (ctx->fp8) {
            // BASELINE configs[3]: the linears named by me_model_config.fp8_linears (default: all four) run on the scaled
            // fp8 MFMA, each fed by a producer that writes MX fp8 -- LayerNorm (qkv, fc1), the attention kernel's own
            // store stage (proj), fc1's epilogue (fc2) -- the others on the 16-bit kernels; attention itself stays f16
            const int mask = ctx->fp8_mask;
            const bool q8 = mask & 1, p8 = mask & 2, f18 = mask & 4, f28 = mask & 8;
            const int windows = W0 + W1 * (fov ? 2 : 1);
            set_ln(b1.ln1_w, b1.ln1_b, b2.ln1_w, b2.ln1_b);
            if (q8) {
                // norm1 as MX fp8: written by the previous block's fp8 fc2 where the LayerNorm rides in its residual epilogue
                if (!xn_is_norm1_fp8)
                    layernorm_fp8_launch(tok, b0.ln1_w, b0.ln1_b, (uint8_t*)xn, xn_s, Rtot, C, ctx->cfg.ln_eps, s, &segs);
                xn_is_norm1_fp8 = false;
                gemm8(xn, xn_s, C, 3 * C, b0, b1, b2, 0);
            } else {
                layernorm_launch(tok, b0.ln1_w, b0.ln1_b, xn, nullptr, Rtot, C, ctx->cfg.ln_eps, ctx->dtype, s, &segs);
                gemm_all(xn, C, b0.qkv_w, b1.qkv_w, b2.qkv_w, b0.qkv_b, b1.qkv_b, b2.qkv_b, 3 * C, qkv, ACT_NONE, C);
            }
            if (p8) {
                // the attention kernel writes the projection's fp8 operand itself (the bytes a separate
                // quantize_f16_to_fp8 pass over its 16-bit output would give: ME_FP8_ATT_SEPARATE=1 runs that pass,
                // the test compares the two)
                static const bool separate = getenv("ME_FP8_ATT_SEPARATE") != nullptr;
                if (separate || heads % 2) {
                    attention_launch(qkv, att, windows, T, heads, ctx->dtype, s, &segs, nullptr, nullptr, 0, true);
                    quantize_f16_to_fp8_launch(att, att8, att_s, Rtot, C, 0, s);
                } else {
                    attention_launch(qkv, att, windows, T, heads, ctx->dtype, s, &segs, att8, att_s, Rtot / 128, true);
                }
                // ... and its residual epilogue norm2 as fc1's fp8 operand (the tall fp8 tile)
                const LnSet ln2{b0.ln2_w, b0.ln2_b, b1.ln2_w, b1.ln2_b, b2.ln2_w, b2.ln2_b};
                xn_is_norm2_fp8 = gemm8(att8, att_s, C, C, b0, b1, b2, 3, false, f18 ? &ln2 : nullptr);
            } else {
                attention_launch(qkv, att, windows, T, heads, ctx->dtype, s, &segs, nullptr, nullptr, 0, true);
                // the 16-bit projection's residual epilogue writes norm2 as fc1's MX fp8 operand (gemm_core.h resid_ln_epilogue)
                const LnSet ln2{b0.ln2_w, b0.ln2_b, b1.ln2_w, b1.ln2_b, b2.ln2_w, b2.ln2_b};
                xn_is_norm2_fp8 = resid_all(att, C, b0.proj_w, b0.proj_b, b0.ls1, b1.proj_w, b1.proj_b, b1.ls1, b2.proj_w, b2.proj_b,
                                            b2.ls1, f18 ? &ln2 : nullptr, true);
            }
            set_ln(b1.ln2_w, b1.ln2_b, b2.ln2_w, b2.ln2_b);
            if (f18) {
                if (!xn_is_norm2_fp8)
                    layernorm_fp8_launch(tok, b0.ln2_w, b0.ln2_b, (uint8_t*)xn, xn_s, Rtot, C, ctx->cfg.ln_eps, s, &segs);
                xn_is_norm2_fp8 = false;
                gemm8(xn, xn_s, C, 4 * C, b0, b1, b2, 1, !f28);
            } else {
                layernorm_launch(tok, b0.ln2_w, b0.ln2_b, xn, nullptr, Rtot, C, ctx->cfg.ln_eps, ctx->dtype, s, &segs);
                gemm_all(xn, C, b0.fc1_w, b1.fc1_w, b2.fc1_w, b0.fc1_b, b1.fc1_b, b2.fc1_b, 4 * C, hid, ACT_GELU);
            }
            if (f28) {
                const uint8_t* h8 = (const uint8_t*)hid;
                if (!f18) {  // a 16-bit fc1 left f16: quantised here as fc2's operand
                    uint8_t* q = (uint8_t*)site_buf(ctx, "vitm.hid8", (size_t)Rtot * 4 * C);
                    quantize_f16_to_fp8_launch(hid, q, hid_s, Rtot, 4 * C, 0, s);
                    h8 = q;
                }
                LnSet next;
                const bool fuse_next = q8 && i + 1 < ctx->cfg.depth;  // the next block's qkv reads norm1 as fp8
                if (fuse_next) {
                    const VitBlockW &n0 = p0.blocks[i + 1], &n1 = p1.blocks[i + 1], &n2 = p2.blocks[i + 1];
                    next = LnSet{n0.ln1_w, n0.ln1_b, n1.ln1_w, n1.ln1_b, n2.ln1_w, n2.ln1_b};
                }
                xn_is_norm1_fp8 = gemm8(h8, hid_s, 4 * C, C, b0, b1, b2, 2, false, fuse_next ? &next : nullptr);
            } else {
                resid_all(hid, 4 * C, b0.fc2_w, b0.fc2_b, b0.ls2, b1.fc2_w, b1.fc2_b, b1.ls2, b2.fc2_w, b2.fc2_b, b2.ls2);
            }
            if (taps.fn) taps.fn(taps.user, i, tok + row_main * C);
            return;
        }
        // norm1: written by the previous block's fc2 launch where the LayerNorm rides in the residual epilogue
        if (!xn_is_norm1) {
            set_ln(b1.ln1_w, b1.ln1_b, b2.ln1_w, b2.ln1_b);
            layernorm_launch(tok, b0.ln1_w, b0.ln1_b, xn, nullptr, Rtot, C, ctx->cfg.ln_eps, ctx->dtype, s, &segs);
        }
        xn_is_norm1 = false;
        gemm_all(xn, C, b0.qkv_w, b1.qkv_w, b2.qkv_w, b0.qkv_b, b1.qkv_b, b2.qkv_b, 3 * C, qkv, ACT_NONE, C);
        attention_launch(qkv, att, W0 + W1 * (fov ? 2 : 1), T, heads, ctx->dtype, s, &segs, nullptr, nullptr, 0, true);
        const LnSet ln2{b0.ln2_w, b0.ln2_b, b1.ln2_w, b1.ln2_b, b2.ln2_w, b2.ln2_b};
        if (!resid_all(att, C, b0.proj_w, b0.proj_b, b0.ls1, b1.proj_w, b1.proj_b, b1.ls1, b2.proj_w, b2.proj_b, b2.ls1, &ln2)) {
            set_ln(b1.ln2_w, b1.ln2_b, b2.ln2_w, b2.ln2_b);
            layernorm_launch(tok, b0.ln2_w, b0.ln2_b, xn, nullptr, Rtot, C, ctx->cfg.ln_eps, ctx->dtype, s, &segs);
        }
        gemm_all(xn, C, b0.fc1_w, b1.fc1_w, b2.fc1_w, b0.fc1_b, b1.fc1_b, b2.fc1_b, 4 * C, hid, ACT_GELU);
        LnSet next;
        const bool has_next = i + 1 < ctx->cfg.depth;
        if (has_next) {
            const VitBlockW &n0 = p0.blocks[i + 1], &n1 = p1.blocks[i + 1], &n2 = p2.blocks[i + 1];
            next = LnSet{n0.ln1_w, n0.ln1_b, n1.ln1_w, n1.ln1_b, n2.ln1_w, n2.ln1_b};
        }
        xn_is_norm1 = resid_all(hid, 4 * C, b0.fc2_w, b0.fc2_b, b0.ls2, b1.fc2_w, b1.fc2_b, b1.ls2, b2.fc2_w, b2.fc2_b, b2.ls2,
                                has_next ? &next : nullptr);
        if (taps.fn) taps.fn(taps.user, i, tok + row_main * C);
    }

    // vit.rs:343 final norm of each ViT; the 16-bit results of segments 0 / 1 / 2
    // want32: also keep the normalised tokens in f32 (the source of split [hi | lo] token maps)
    void finish(bool want32) {
        set_ln(p1.norm_w, p1.norm_b, p2.norm_w, p2.norm_b);
        fin32 = want32 ? (float*)site_buf(ctx, "vitm.final32", (size_t)Rtot * ctx->C() * 4) : nullptr;
        layernorm_launch(tok, p0.norm_w, p0.norm_b, fin16, fin32, Rtot, ctx->C(), ctx->cfg.ln_eps, ctx->dtype, s,
                         &segs);
    }
    // seg: 0 = patch encoder, 1 = image encoder, 2 = FOV encoder (the ViT, not the physical position)
    int64_t seg_row(int seg) const { return seg == 0 ? row_main : (seg == 1 ? row_img : row_fov); }
    void* final16(int seg) const { return fin16 + seg_row(seg) * ctx->C() * 2; }
    const float* final32(int seg) const { return fin32 + seg_row(seg) * ctx->C(); }
};
}  // namespace

void vit_forward(me_ctx* ctx, int which, const void* patches16, int W, const VitTaps& taps,
                 void* final16, float* final32, const std::string& tag, hipStream_t s) {
    VitRun run(ctx, which, patches16, W, taps, tag, s);
    for (int i = 0; i < ctx->cfg.depth; ++i) run.block(i);
    run.finish(final16, final32);
}

// encoder.rs:218-335 DepthProEncoder::forward_encodings
void stage_encoder(me_ctx* ctx, const float* img32, int B, bool with_fov) {
    stage_encoder_trunk(ctx, img32, B, with_fov);
    stage_encoder_latents(ctx, B);
}

// encoder.rs:307-309: the two latent upsample chains (taps of blocks 5 and 11 -> encodings 0 and 1).  Nothing but decoder
// levels 1 and 0 reads their results, and their ConvTranspose launches are bound by HBM writes: extract_depth_impl runs
// them beside the low-resolution decoder levels (api.hip).
void stage_encoder_latents(me_ctx* ctx, int B) {
    hipStream_t s = ctx->stream;
    const me_model_config& c = ctx->cfg;
    const int g = ctx->g(), dec = c.dec_dim, e0 = c.enc_dims[0];
    const bool sp_dec = ctx->split(SPLIT_DEC_CONVS);
    const size_t wide_dec = sp_dec ? 2 : 1;
    const int side0 = 4 * g, H0 = 32 * g, H1 = 16 * g;
    const void* lat0 = ctx->bufs.at("enc.lat0").p;
    const void* lat1 = ctx->bufs.at("enc.lat1").p;
    void* enc1 = site_buf(ctx, "enc1.16b", bordered_bytes(B, H1, H1, e0) * wide_dec);
    run_upsample(ctx, "up_latent1", lat1, B, side0, ctx->w.up_latent1, nullptr, enc1, true, 0,
                 ACT_NONE, s, sp_dec);
    float* enc0_32 = (float*)site_buf(ctx, "enc0.f32", (size_t)B * H0 * H0 * dec * 4);
    void* enc0_r16 = site_buf(ctx, "enc0.r16b", bordered_bytes(B, H0, H0, dec));
    run_upsample(ctx, "up_latent0", lat0, B, side0, ctx->w.up_latent0, enc0_32, enc0_r16, true, 0,
                 ACT_RELU, s);
}

void stage_encoder_trunk(me_ctx* ctx, const float* img32, int B, bool fov_async /* = with_fov */) {
    hipStream_t s = ctx->stream;
    const me_model_config& c = ctx->cfg;
    const int g = ctx->g(), S = ctx->S(), C = ctx->C(), P = ctx->P(), T = ctx->T();
    const int dec = c.dec_dim, e1 = c.enc_dims[1], e2 = c.enc_dims[2], e3 = c.enc_dims[3];
    const bool sp = ctx->split(SPLIT_UPSAMPLE);  // merged token maps and upsample intermediates as [hi | lo]
    const bool sp_dec = ctx->split(SPLIT_DEC_CONVS);  // encodings 1..4 as [hi | lo] for decoder.convs
    const size_t wide = sp ? 2 : 1, wide_dec = sp_dec ? 2 : 1;
    report(ctx, 0.0f, "creating image pyramid");
    // encoder.rs:125-140 create_pyramid (x0 itself only changes type)
    void* x0 = site_buf(ctx, "enc.x0", (size_t)B * 3 * S * S * 2);
    void* x1 = site_buf(ctx, "enc.x1", (size_t)B * 3 * (S / 2) * (S / 2) * 2);
    void* x2 = site_buf(ctx, "enc.x2", (size_t)B * 3 * (S / 4) * (S / 4) * 2);
    cast_f32_to_16_launch(img32, x0, (int64_t)B * 3 * S * S, ctx->dtype, s);
    bilinear_launch(img32, x1, 3 * B, S, S / 2, c.align_corners, ctx->dtype, s);
    bilinear_launch(img32, x2, 3 * B, S, S / 4, c.align_corners, ctx->dtype, s);
    void* xg = site_buf(ctx, "enc.xg", (size_t)B * g * g * C * 2 * wide);
    // encoder.rs:238-250 split + cat, vit.rs:210-223 patch embed im2col
    report(ctx, 0.02f, "preparing image patches");
    void* patches = site_buf(ctx, "enc.patches", (size_t)B * 35 * P * 768 * 2);
    patchify_launch(x0, x1, x2, patches, B, g, ctx->dtype, s);

    report(ctx, 0.03f, "encoding patches");
    const int side0 = 4 * g, side1 = 2 * g;
    void* lat0 = site_buf(ctx, "enc.lat0", (size_t)B * side0 * side0 * C * 2 * wide);
    void* lat1 = site_buf(ctx, "enc.lat1", (size_t)B * side0 * side0 * C * 2 * wide);
    TapCtx tc{ctx, B, lat0, lat1, s};
    VitTaps taps;
    taps.fn = tap_fn, taps.user = &tc;
    // encoder.rs:298-303 image encoder on the 1/4 image and (fov.rs:57-63) the FOV encoder: both depend only
    // on x2; the three ViTs run as one row space on the launch stream (MergedVit above)
    void* patches2 = site_buf(ctx, "enc.patches2", (size_t)B * P * 768 * 2);
    patchify_windows_launch(x2, patches2, B, g, ctx->dtype, s);  // both small ViTs embed the same patches
    if (fov_async) report(ctx, 0.03f, "encoding fov");
    MergedVit run(ctx, B, fov_async, patches, patches2, patches2, taps, s);
    for (int i = 0; i < c.depth; ++i) run.block(i);
    run.finish(sp);
    // the final tokens feed merge: as 16-bit rows, or (split operands) as f32 rows that merge splits
    const float* tok32 = sp ? run.final32(0) : nullptr;
    const void* tok16 = sp ? nullptr : run.final16(0);
    merge_launch(sp ? run.final32(1) : nullptr, sp ? nullptr : run.final16(1), xg, B, 1, 0, 1, 0, g, C, ctx->dtype,
                 s, sp);
    if (fov_async) {
        float* lin32 = (float*)site_buf(ctx, "fov.lin", (size_t)B * T * (dec / 2) * 4);
        linear(ctx, run.final16(2), (int64_t)B * T, C, ctx->w.fov_lin_w, dec / 2, ctx->w.fov_lin_b, nullptr, lin32,
               dec / 2, ACT_NONE, s);
    }
    report(ctx, 0.55f, "reshaping patch encodings");
    // encoder.rs:263,285-294: split_with_sizes + merge
    void* x0f = site_buf(ctx, "enc.x0f", (size_t)B * side0 * side0 * C * 2 * wide);
    void* x1f = site_buf(ctx, "enc.x1f", (size_t)B * side1 * side1 * C * 2 * wide);
    void* x2f = site_buf(ctx, "enc.x2f", (size_t)B * g * g * C * 2 * wide);
    merge_launch(tok32, tok16, x0f, B, 35, 0, 5, g / 8, g, C, ctx->dtype, s, sp);
    merge_launch(tok32, tok16, x1f, B, 35, 25, 3, g / 4, g, C, ctx->dtype, s, sp);
    merge_launch(tok32, tok16, x2f, B, 35, 34, 1, 0, g, C, ctx->dtype, s, sp);

    report(ctx, 0.7f, "encoding features");
    // encoder.rs:307-316
    const int H2 = 8 * g, H3 = 4 * g, H4 = 2 * g;   // (encodings 0 and 1: stage_encoder_latents)
    void* enc2 = site_buf(ctx, "enc2.16b", bordered_bytes(B, H2, H2, e1) * wide_dec);
    run_upsample(ctx, "up0", x0f, B, side0, ctx->w.up0, nullptr, enc2, true, 0, ACT_NONE, s, sp_dec);
    void* enc3 = site_buf(ctx, "enc3.16b", bordered_bytes(B, H3, H3, e2) * wide_dec);
    run_upsample(ctx, "up1", x1f, B, side1, ctx->w.up1, nullptr, enc3, true, 0, ACT_NONE, s, sp_dec);
    // encoder.rs:316-325: upsample2, upsample_lowres, cat on channels, fuse_lowres.  Split: a pixel of `cat`
    // is [up2 hi | lowres hi | up2 lo | lowres lo], i.e. [hi(2 e3) | lo(2 e3)] against fuse_lowres' [W | W]
    char* cat = (char*)site_buf(ctx, "enc.cat", (size_t)B * H4 * H4 * 2 * e3 * 2 * wide);
    const int64_t cat_stride = 2 * e3 * (int64_t)wide;
    run_upsample(ctx, "up2", x2f, B, g, ctx->w.up2, nullptr, cat, false, cat_stride, ACT_NONE, s, sp, 2 * e3);
    report(ctx, 0.9f, "upsampling lowres");
    convt(ctx, xg, B, g, g, C, ctx->w.up_lowres_w, e3, ctx->w.up_lowres_b, nullptr, cat + (size_t)e3 * 2,
          false, cat_stride, ACT_NONE, s, sp, sp, 2 * e3);
    report(ctx, 0.95f, "fusing lowres");
    void* enc4 = site_buf(ctx, "enc4.16b", bordered_bytes(B, H4, H4, e3) * wide_dec);
    {
        GemmParams p = base_params();
        const int K = 2 * e3 * (int)wide;
        p.M = B * H4 * H4, p.N = e3, p.K = K, p.flop_k = 2 * e3, p.A = cat, p.lda = K, p.W = ctx->w.fuse_w;
        p.bias = ctx->w.fuse_b, p.out16 = enc4, p.out16_border = 1, p.ldc = e3;
        p.out_H = H4, p.out_W = H4;
        if (sp_dec) set_out16_split(p, e3);
        gemm_launch(p, A_PLAIN, EPI_STORE, ctx->dtype, s);
    }
}

// the head behind head[0] as one launch on the half-resolution map (stage_head): needs the un-split chain and the 128-channel
// halo tile's shapes; ME_HEAD_COMPOSED=0: the three launches
static bool head_composed_applies(const me_ctx* ctx, int B) {
    static const bool composed_on = !(getenv("ME_HEAD_COMPOSED") && atoi(getenv("ME_HEAD_COMPOSED")) == 0);
    const int dec = ctx->cfg.dec_dim, Hh = ctx->S() / 2;
    return composed_on && !ctx->split(SPLIT_HEAD) && ctx->w.head_fused_w && Hh % 16 == 0 && (dec / 2) % 64 == 0 &&
           ((int64_t)B * Hh * Hh) % 256 == 0;
}

// decoder.rs:153-208 MultiresConvDecoder::forward (+ :84-102 FeatureFusionBlock, :35-44 RCU)
void stage_decoder(me_ctx* ctx, int B, bool want_features32) { stage_decoder_levels(ctx, B, want_features32, 4, 0); }

void stage_decoder_levels(me_ctx* ctx, int B, bool want_features32, int first, int last) {
    hipStream_t s = ctx->stream;
    const me_model_config& c = ctx->cfg;
    const int g = ctx->g(), dec = c.dec_dim;
    const int H[5] = {32 * g, 16 * g, 8 * g, 4 * g, 2 * g};
    const int Cin[5] = {dec, c.enc_dims[0], c.enc_dims[1], c.enc_dims[2], c.enc_dims[3]};
    const char* enc_names[5] = {"enc0.r16b", "enc1.16b", "enc2.16b", "enc3.16b", "enc4.16b"};

    const bool sp_dec = ctx->split(SPLIT_DEC_CONVS), spf = ctx->split(SPLIT_FUSION_OUT),
               sp_head = ctx->split(SPLIT_HEAD);
    // `features` carried between levels (f32 residual path): the level above left them in its ".feat.f32" buffer
    float* feat32 = first == 4 ? nullptr : (float*)ctx->bufs.at("dec" + std::to_string(first + 1) + ".feat.f32").p;
    for (int i = first; i >= last; --i) {
        report(ctx, (4 - i) / 5.0f, i == 4 ? "decoding initial block" : "decoding blocks");
        const std::string L = "dec" + std::to_string(i);
        const int h = H[i];
        const size_t n32 = (size_t)B * h * h * dec * 4, nb = bordered_bytes(B, h, h, dec);
        const FusionW& fw = ctx->w.fusions[i];
        const void* enc = ctx->bufs.at(enc_names[i]).p;

        // features_i = convs[i-1](encoding) (3x3, no bias); level 0 uses the encoding as is
        float* x1_32;
        void* x1_r16;
        if (i == 0) {
            x1_32 = (float*)ctx->bufs.at("enc0.f32").p;
            x1_r16 = (void*)enc;
        } else {
            // level 4's conv output is `lowres_features` (decoder.rs:178), kept for the FOV head
            x1_32 = (float*)site_buf(ctx, i == 4 ? std::string("lowres.f32") : L + ".x1.f32", n32);
            x1_r16 = site_buf(ctx, L + ".x1.r16b", nb);
            ConvOut o;
            o.out32 = x1_32, o.out16 = x1_r16, o.border16 = true, o.act = ACT_RELU;
            conv(ctx, enc, B, h, h, Cin[i], ctx->w.dec_convs[i], dec, 3, 1, nullptr, o, s, sp_dec);
        }

        float* out32;
        void* out_r16;
        if (i == 4) {
            // decoder.rs:171-183: lowres_features = features.clone(); fusions.last()(features, None)
            out32 = x1_32;
            out_r16 = x1_r16;
        } else {
            // out = x0 + resnet1(x1) = x0 + x1 + conv2(relu(conv1(relu(x1))))
            void* t_r16 = site_buf(ctx, L + ".t1.r16b", nb);
            ConvOut o1;
            o1.out16 = t_r16, o1.border16 = true, o1.act = ACT_RELU;
            conv(ctx, x1_r16, B, h, h, dec, fw.resnet1.w[0], dec, 3, 1, fw.resnet1.b[0], o1, s);
            out32 = (float*)site_buf(ctx, L + ".out.f32", n32);
            out_r16 = site_buf(ctx, L + ".out.r16b", nb);
            ConvOut o2;
            o2.out32 = out32, o2.out16 = out_r16, o2.border16 = true, o2.act = ACT_RELU;
            o2.res32 = x1_32, o2.res32b = feat32;
            conv(ctx, t_r16, B, h, h, dec, fw.resnet1.w[1], dec, 3, 1, fw.resnet1.b[1], o2, s);
        }
        // resnet2
        void* t2_r16 = site_buf(ctx, L + ".t2.r16b", nb);
        ConvOut o3;
        o3.out16 = t2_r16, o3.border16 = true, o3.act = ACT_RELU;
        conv(ctx, out_r16, B, h, h, dec, fw.resnet2.w[0], dec, 3, 1, fw.resnet2.b[0], o3, s);
        // the operands of deconv and out_conv have no residual path beside them: [hi | lo] under SPLIT_FUSION_OUT,
        // [hi | lo | hi] where the two are composed into one ConvTranspose (FusionW::fused_w)
        const bool fused = spf && fw.deconv && fw.fused_w;
        const size_t wf = fused ? 3 : (spf ? 2 : 1);
        if (i == 0) {
            // decoder.rs:101 out_conv + mod.rs:326 head[0] as ONE convolution (weights.hip compose_features): this level's last
            // convolution writes out_conv's INPUT as head[0]'s zero-bordered operand, the 1x1 launch and the feature map between
            // them (0.9 GB of traffic at 768 x 768) go.  Not when the caller wants the features themselves, nor under SPLIT_HEAD;
            // ME_FEAT_COMPOSED=0: the two layers as they are.
            static const bool feat_composed_on = !(getenv("ME_FEAT_COMPOSED") && atoi(getenv("ME_FEAT_COMPOSED")) == 0);
            ctx->features_pre = feat_composed_on && !want_features32 && head_composed_applies(ctx, B) && ctx->w.feat_fused_w;
            if (ctx->features_pre) {
                ConvOut o4;
                o4.out16 = site_buf(ctx, "features.16b", bordered_bytes(B, h, h, dec)), o4.border16 = true, o4.res32 = out32;
                conv(ctx, t2_r16, B, h, h, dec, fw.resnet2.w[1], dec, 3, 1, fw.resnet2.b[1], o4, s);
                continue;
            }
        }
        void* v16 = site_buf(ctx, L + ".v16", (size_t)B * h * h * dec * 2 * wf);
        ConvOut o4;
        o4.out16 = v16, o4.res32 = out32, o4.split16 = spf, o4.triple16 = fused;
        conv(ctx, t2_r16, B, h, h, dec, fw.resnet2.w[1], dec, 3, 1, fw.resnet2.b[1], o4, s);
        if (fused) {
            // out_conv(deconv(v)) as ONE launch: f32 features of the next level straight from the pixel shuffle
            const int64_t Mo = (int64_t)B * 4 * h * h;
            feat32 = (float*)site_buf(ctx, L + ".feat.f32", (size_t)Mo * dec * 4);
            convt(ctx, v16, B, h, h, dec, fw.fused_w, dec, fw.out_b, feat32, nullptr, false, 0, ACT_NONE, s, false, false,
                  0, 3);
            continue;
        }
        // deconv (levels 1-4) then out_conv 1x1 (+bias)
        const void* pre = v16;
        int ho = h;
        if (fw.deconv) {
            void* d16 = site_buf(ctx, L + ".d16", (size_t)B * 4 * h * h * dec * 2 * wf);
            convt(ctx, v16, B, h, h, dec, fw.deconv, dec, nullptr, nullptr, d16, false, 0, ACT_NONE, s, spf, spf);
            pre = d16, ho = 2 * h;
        }
        const int64_t Mo = (int64_t)B * ho * ho;
        if (i > 0) {
            feat32 = (float*)site_buf(ctx, L + ".feat.f32", (size_t)Mo * dec * 4);
            linear(ctx, pre, Mo, dec, fw.out_w, dec, fw.out_b, nullptr, feat32, dec, ACT_NONE, s, spf);
        } else {
            // final features: 16-bit zero-bordered operand of head[0]; f32 copy for the ABI
            float* f32 = want_features32 ? (float*)site_buf(ctx, "features.f32", (size_t)Mo * dec * 4)
                                         : nullptr;
            void* f16b = site_buf(ctx, "features.16b", bordered_bytes(B, ho, ho, dec) * (sp_head ? 2 : 1));
            GemmParams p = base_params();
            const int K = dec * (int)wf;
            p.M = (int)Mo, p.N = dec, p.K = K, p.flop_k = dec, p.A = pre, p.lda = K, p.W = fw.out_w;
            p.bias = fw.out_b, p.out16 = f16b, p.out16_border = 1, p.out32 = f32, p.ldc = dec;
            p.out_H = ho, p.out_W = ho;
            if (sp_head) set_out16_split(p, dec);
            gemm_launch(p, A_PLAIN, EPI_STORE, ctx->dtype, s);
        }
    }
}

// mod.rs:323-333 head convs + ReLUs, mod.rs:361-362 div_scalar + clamp fused into the last kernel
// pre_image: "features.16b" holds the input of the last fusion block's out_conv instead of its output (ctx->features_pre)
void stage_head(me_ctx* ctx, int B, const float* f_norm_dev, bool clamp, float* depth_dev, bool pre_image) {
    hipStream_t s = ctx->stream;
    const me_model_config& c = ctx->cfg;
    const int dec = c.dec_dim, S = ctx->S(), Hh = S / 2;
    report(ctx, 0.0f, "forwarding head");
    // mod.rs:323-333 is one chain without a residual path: its operands are [hi | lo] under SPLIT_HEAD
    const bool sp = ctx->split(SPLIT_HEAD);
    const size_t wide = sp ? 2 : 1;
    const void* f16b = ctx->bufs.at("features.16b").p;
    // mod.rs:326-333 behind head[0]: ConvTranspose -> conv3x3 -> ReLU -> conv1x1 -> ReLU as ONE launch on the half-resolution map,
    // the two linear layers composed at load time (weights.hip compose_head): the [B, 128, 1536, 1536] tensor between them (605 MB
    // written and read back) never exists.  Needs the un-split chain and the 128-channel halo tile's shapes; ME_HEAD_COMPOSED=0:
    // the three launches below.
    ME_CHECK(!pre_image || head_composed_applies(ctx, B), ME_ERR_BAD_ARG, "head: composed features without the composed head");
    if (head_composed_applies(ctx, B)) {
        void* h0b = site_buf(ctx, "head.h0b", bordered_bytes(B, Hh, Hh, dec / 2));
        ConvOut o;
        o.out16 = h0b, o.border16 = true;
        if (pre_image) {
            o.tap_bias = ctx->w.feat_fused_b + dec / 2;
            conv(ctx, f16b, B, Hh, Hh, dec, ctx->w.feat_fused_w, dec / 2, 3, 1, ctx->w.feat_fused_b, o, s, false);
        } else {
            conv(ctx, f16b, B, Hh, Hh, dec, ctx->w.head0_w, dec / 2, 3, 1, ctx->w.head0_b, o, s, false);
        }
        GemmParams p = base_params();
        p.M = B * Hh * Hh, p.N = 128, p.K = 9 * (dec / 2);
        p.A = h0b, p.in_Hp = Hh + 2, p.in_Wp = Hh + 2, p.Cin = dec / 2, p.out_H = Hh, p.out_W = Hh;
        p.KH = 3, p.KW = 3, p.stride = 1, p.W = ctx->w.head_fused_w, p.bias = ctx->w.head_fused_b;
        p.tap_bias = ctx->w.head_fused_b + 32;
        p.w2 = ctx->w.head4_w, p.b2 = ctx->w.head4_b, p.f_norm = f_norm_dev;
        p.pixels_per_image = S * S, p.out32 = depth_dev;
        p.clamp_lo = clamp ? 1e-4f : -INFINITY, p.clamp_hi = clamp ? 1e4f : INFINITY;
        head_composed_launch(p, ctx->dtype, s);
        return;
    }
    void* h0 = site_buf(ctx, "head.h0", (size_t)B * Hh * Hh * (dec / 2) * 2 * wide);
    ConvOut o;
    o.out16 = h0, o.split16 = sp;
    conv(ctx, f16b, B, Hh, Hh, dec, ctx->w.head0_w, dec / 2, 3, 1, ctx->w.head0_b, o, s, sp);
    void* h1 = site_buf(ctx, "head.h1b", bordered_bytes(B, S, S, dec / 2) * wide);
    convt(ctx, h0, B, Hh, Hh, dec / 2, ctx->w.head1_w, dec / 2, ctx->w.head1_b, nullptr, h1, true, 0,
          ACT_NONE, s, sp, sp);
    GemmParams p = base_params();
    const int hc = (dec / 2) * (int)wide;
    p.M = B * S * S, p.N = c.head_dims[0], p.K = 9 * hc, p.flop_k = 9 * (dec / 2);
    p.A = h1, p.in_Hp = S + 2, p.in_Wp = S + 2, p.Cin = hc, p.out_H = S, p.out_W = S;
    p.KH = 3, p.KW = 3, p.stride = 1, p.W = ctx->w.head2_w, p.bias = ctx->w.head2_b;
    p.w2 = ctx->w.head4_w, p.b2 = ctx->w.head4_b, p.f_norm = f_norm_dev;
    p.pixels_per_image = S * S, p.out32 = depth_dev;
    if (clamp) p.clamp_lo = 1e-4f, p.clamp_hi = 1e4f;
    gemm_launch(p, A_CONV, EPI_HEAD_FINAL, ctx->dtype, s);
}

// fov.rs:40-63: the FOV encoder (third ViT-L on "enc.x2", the 1/4 image) and its Linear
void stage_fov_vit(me_ctx* ctx, int B, hipStream_t s) {
    const int g = ctx->g(), C = ctx->C(), P = ctx->P(), T = ctx->T(), dec = ctx->cfg.dec_dim;
    report(ctx, 0.0f, "encoding fov");
    const void* x2 = ctx->bufs.at("enc.x2").p;
    void* patches = site_buf(ctx, "fov.patches", (size_t)B * P * 768 * 2);
    patchify_windows_launch(x2, patches, B, g, ctx->dtype, s);
    void* tok16 = site_buf(ctx, "fov.tok16", (size_t)B * T * C * 2);
    vit_forward(ctx, ME_VIT_FOV_ENCODER, patches, B, VitTaps(), tok16, nullptr, "vit.fov", s);
    float* lin32 = (float*)site_buf(ctx, "fov.lin", (size_t)B * T * (dec / 2) * 4);
    linear(ctx, tok16, (int64_t)B * T, C, ctx->w.fov_lin_w, dec / 2, ctx->w.fov_lin_b, nullptr, lin32,
           dec / 2, ACT_NONE, s);
}

// fov.rs:66-88: the convolutional tail; needs "fov.lin" (stage_fov_vit) and "lowres.f32" (decoder)
void stage_fov_tail(me_ctx* ctx, int B, float* fov_deg_dev) {
    hipStream_t s = ctx->stream;
    const me_model_config& c = ctx->cfg;
    const int g = ctx->g(), P = ctx->P(), T = ctx->T(), dec = c.dec_dim;
    report(ctx, 0.85f, "fov lowres");
    const float* lin32 = (const float*)ctx->bufs.at("fov.lin").p;
    // fov.rs:70-74: relu(downsample[0](lowres)) + reshaped tokens
    const float* low32 = (const float*)ctx->bufs.at("lowres.f32").p;
    void* low16b = site_buf(ctx, "fov.low16b", bordered_bytes(B, 2 * g, 2 * g, dec));
    nhwc32_to_16b_launch(low32, low16b, B, 2 * g, 2 * g, dec, 0, ctx->dtype, s);
    float* fl32 = (float*)site_buf(ctx, "fov.down", (size_t)B * P * (dec / 2) * 4);
    ConvOut o0;
    o0.out32 = fl32, o0.act = ACT_RELU, o0.act16_only = false;
    conv(ctx, low16b, B, 2 * g, 2 * g, dec, ctx->w.fov_down_w, dec / 2, 3, 2, ctx->w.fov_down_b, o0, s);
    void* fx = site_buf(ctx, "fov.x16b", bordered_bytes(B, g, g, dec / 2));
    fov_add_relu_launch(lin32, fl32, fx, B, g, dec / 2, T, ctx->dtype, s);
    // fov.rs:77-85 head
    void* f1 = site_buf(ctx, "fov.h0b", bordered_bytes(B, g / 2, g / 2, dec / 4));
    ConvOut o1;
    o1.out16 = f1, o1.border16 = true, o1.act = ACT_RELU;
    conv(ctx, fx, B, g, g, dec / 2, ctx->w.fov_h0_w, dec / 4, 3, 2, ctx->w.fov_h0_b, o1, s);
    void* f2 = site_buf(ctx, "fov.h2", (size_t)B * (g / 4) * (g / 4) * (dec / 8) * 2);
    ConvOut o2;
    o2.out16 = f2, o2.act = ACT_RELU;
    conv(ctx, f1, B, g / 2, g / 2, dec / 4, ctx->w.fov_h2_w, dec / 8, 3, 2, ctx->w.fov_h2_b, o2, s);
    float* fnorm = (float*)site_buf(ctx, "f_norm", (size_t)B * 4);
    fov_final_launch(f2, ctx->w.fov_h4_w, ctx->w.fov_h4_b, fov_deg_dev, fnorm, B, g / 4, dec / 8,
                     ctx->dtype, s);
}

}  // namespace me
