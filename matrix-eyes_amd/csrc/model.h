// Context, packed weights and stage functions of the Depth Pro forward pass.
#pragma once
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

namespace me {

// How a checkpoint tensor (PyTorch layout) is packed into the device arena.
enum PackKind : int32_t {
    PK_VEC_F32 = 0,  // any shape -> flat f32 (biases, LayerNorm, LayerScale, cls/pos)
    PK_MAT_16 = 1,   // [N][K...] row-major -> 16-bit [N][K]  (Linear, 1x1 conv, patch embed)
    PK_CONV_16 = 2,  // Conv2d [Cout][Cin][kh][kw] -> 16-bit [Cout][kh*kw][Cin]
    PK_CONVT_16 = 3, // ConvTranspose2d [Cin][Cout][2][2] -> 16-bit [(dy*2+dx)*Cout + co][Cin]
    PK_CONVK_F32 = 4 // Conv2d [1][Cin][k][k] -> f32 [k][k][Cin]  (fov.head.4)
};

struct WeightSlot {
    std::string name;
    std::vector<int64_t> dims;
    PackKind kind;
    size_t offset = 0, bytes = 0;
    bool loaded = false;
    // the K axis is stored twice, [W | W] per row / per tap: the consumer reads a split [hi | lo] activation
    // operand of twice the depth (pipeline.hip, "split operands")
    bool dup = false;
    int64_t numel() const {
        int64_t n = 1;
        for (auto d : dims) n *= d;
        return n;
    }
};

struct VitBlockW {
    const float *ln1_w, *ln1_b, *qkv_b, *proj_b, *ls1, *ln2_w, *ln2_b, *fc1_b, *fc2_b, *ls2;
    const void *qkv_w, *proj_w, *fc1_w, *fc2_w;
    // ME_DTYPE_FP8: MX fp8 copies of the four linears (e4m3 bytes [N][K] + e8m0 block scales in the weight
    // layout of mx_fp8.h), quantised on the device from the 16-bit arena when the weights are finalized
    const uint8_t *qkv_w8 = nullptr, *qkv_ws = nullptr, *fc1_w8 = nullptr, *fc1_ws = nullptr, *fc2_w8 = nullptr,
                  *fc2_ws = nullptr, *proj_w8 = nullptr, *proj_ws = nullptr;
};
struct VitW {
    const void* patch_w;
    const float *patch_b, *cls, *pos, *norm_w, *norm_b;
    std::vector<VitBlockW> blocks;
};
struct UpsampleW {
    const void* conv;               // 1x1 [dim_int][C]
    std::vector<const void*> convt; // packed [4*Cout][Cin]
    std::vector<int> cin, cout;
    int dim_int;
};
struct RcuW {
    const void* w[2];
    const float* b[2];
};
struct FusionW {
    RcuW resnet1, resnet2;
    const void* deconv;  // null at level 0
    const void* out_w;
    const float* out_b;
    // SPLIT_FUSION_OUT, levels 1-4: out_conv o deconv composed into ONE ConvTranspose (both are linear and
    // nothing sits between them, decoder.rs:95-101), packed [(dy,dx,co)][W_hi | W_hi | W_lo] against [hi | lo | hi]
    // activations; null when the two run as separate launches
    const void* fused_w = nullptr;
};
struct ModelW {
    VitW vit[3];
    UpsampleW up_latent0, up_latent1, up0, up1, up2;
    const void* up_lowres_w;
    const float* up_lowres_b;
    const void* fuse_w;
    const float* fuse_b;
    const void* dec_convs[5];  // [i] for level i (null for level 0)
    FusionW fusions[5];
    const void *head0_w, *head1_w, *head2_w;
    // derived (weights.hip compose_head): head.1 (ConvTranspose 2x2 s2) and head.2 (conv 3x3) as ONE 3x3 convolution on the
    // half-resolution map with 4 x 32 output channels (output phase (dy, dx) x channel), [128][9][Cmid] 16-bit; head_fused_b:
    // f32 [32] bias of interior pixels, then [9][32] the share of each 3x3 tap in it (taken out again where the tap falls
    // outside the full-resolution image).  Null when the composition does not apply (SPLIT_HEAD).
    const void* head_fused_w = nullptr;
    const float* head_fused_b = nullptr;
    // derived (weights.hip compose_features): fusions[0].out_conv and head.0 as ONE 3x3 convolution of out_conv's input,
    // [dec / 2][9][dec] 16-bit; feat_fused_b: f32 [dec / 2] bias + [9][dec / 2] per-tap shares
    const void* feat_fused_w = nullptr;
    const float* feat_fused_b = nullptr;
    const float *head0_b, *head1_b, *head2_b, *head4_w, *head4_b;
    const void *fov_lin_w, *fov_down_w, *fov_h0_w, *fov_h2_w;
    const float *fov_lin_b, *fov_down_b, *fov_h0_b, *fov_h2_b, *fov_h4_w, *fov_h4_b;
};

// Split operands.  Every 16-bit MFMA operand costs one rounding of 2^-11 relative; on the un-diluted chain
// tokens -> upsample convs -> ... -> fusion out_conv -> head (no residual path beside it) those roundings are
// most of the depth error against the fp32 reference.  A stage listed here stores its 16-bit outputs as
// hi = T(v), lo = T(v - hi) in twice the channels, and its consumer runs the same kernel over K' = 2K against
// [W | W]: the product sees v to ~2^-22 at twice the MFMA work of that (small) layer.
enum SplitStage : int32_t {
    SPLIT_UPSAMPLE = 1,  // encoder.rs:307-325 upsample_* / upsample_lowres / fuse_lowres (A operands from merge)
    SPLIT_FUSION_OUT = 2,  // decoder.rs:95-101 deconv + out_conv of every fusion block
    SPLIT_HEAD = 4,        // mod.rs:323-333 head convs
    SPLIT_DEC_CONVS = 8    // decoder.rs:189-195 convs[i-1] on the encodings
};

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

}  // namespace me

// The opaque C handle.
struct me_ctx {
    int device = 0;
    int32_t dtype = ME_DTYPE_F16;
    // ME_DTYPE_FP8 (BASELINE configs[3]): `dtype` is ME_DTYPE_F16 for every 16-bit operand, and the qkv / fc1 /
    // fc2 linears of the three ViTs run on MX block-scaled fp8 (gemm_fp8.hip)
    bool fp8 = false;
    // which linears of a block run on fp8 in an ME_DTYPE_FP8 context: 1 = qkv, 2 = proj, 4 = fc1, 8 = fc2
    // (me_model_config.fp8_linears; 0 there means all four)
    int32_t fp8_mask = 0;
    char* arena8 = nullptr;  // the fp8 weight copies (derived data: rebuilt after finalize / adopt / broadcast)
    size_t arena8_bytes = 0;
    // Stages whose 16-bit activation operands are carried as hi + lo pairs (me::SplitStage bits)
    int32_t split_mask = 0;
    bool split(int stage_bit) const { return (split_mask & stage_bit) != 0; }
    me_model_config cfg;
    hipStream_t own_stream = nullptr, stream = nullptr;
    // The side branch of extract_depth (api.hip extract_depth_impl): the latency-bound low-resolution decoder levels and
    // the FOV tail run on `side_stream` beside the bandwidth-bound ConvTranspose chains of the encoder's two latents,
    // forked and joined by events (captured into the step's hipGraph like any other dependency).  Opt-in (ME_OVERLAP_TAIL=1); a
    // progress callback or per-kernel timing keep the step on one stream.
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool overlap_tail = true;
    // persistent GEMM launches issued while the side branch is in flight leave this many of the 256 CUs to it
    // (GemmParams::grid_cap); 0: no cap
    int32_t grid_cap = 0;
    std::string last_error;
    me_progress_fn progress = nullptr;
    void* progress_user = nullptr;
    // SplitProgressListener (mod.rs:374-414): a stage's positions in [0,1] are mapped into this range
    float prog_lo = 0.0f, prog_hi = 1.0f;

    // weights
    std::vector<me::WeightSlot> slots;
    std::map<std::string, int> slot_by_name;
    char* arena = nullptr;
    size_t arena_bytes = 0;
    bool finalized = false;
    me::ModelW w;
    std::vector<std::string> unused_weights;  // checkpoint keys me_load_checkpoint_pt skipped
    // derived weights (composed deconv + out_conv per fusion level): arena offsets, and the host copies of their
    // two factors kept from me_load_weight until me_weights_finalize composes them
    size_t fused_off[5] = {0, 0, 0, 0, 0};
    size_t head_fused_off = 0;   // 0: not composed
    size_t feat_fused_off = 0;   // 0: not composed
    bool features_pre = false;   // "features.16b" holds out_conv's INPUT (stage_decoder_levels with the composed head[0])
    std::map<std::string, std::vector<float>> factor_keep;

    // me_status_flags: one device word the kernels OR bits into (ME_STATUS_OVERFLOW_16BIT: an f16 operand store
    // met a magnitude beyond 65504, common.h raise_overflow16)
    unsigned* status_dev = nullptr;
    // Pinned host mirror of the status word: every step whose result stays on the device ends with an asynchronous copy
    // of the word into it, so the NEXT entry into the library sees what earlier asynchronous steps raised without
    // synchronising anything (api.hip check_pending_status).
    volatile unsigned* status_host = nullptr;
    // The fused residual + LayerNorm launch waits on sibling workgroups inside the launch (gemm_core.h
    // resid_ln_epilogue), which needs them co-resident.  When a step reports ME_STATUS_SYNC_TIMEOUT (a second tenant on
    // the device, a CU mask the runtime does not report) the context latches this, the step is run again with the
    // stand-alone LayerNorm launches (bit for bit the ME_LN_FUSE=0 result), and fusion stays off for the context.
    bool ln_fuse_off = false;
    int32_t ln_fallbacks = 0;     // steps re-run because of it (me_ln_fusion_state)
    std::string ln_fuse_note;     // logged once: why fusion went off

    // Write-behind of OBJ files (me_ctx_set_write_behind): the text of mesh call i is written to its file by a host
    // thread while the caller goes on to image i + 1.  One slot = a pinned host staging buffer (the OBJ text's D2H
    // copy, mesh_writer.hip) + the write that reads it; slot 0 alone serves the synchronous form.  Slots are used in
    // turn, so at most `write_behind` writes are in flight.  A failed write is reported by the call that next waits
    // for it (a later me_output_mesh reusing the slot, or me_output_flush).
    int write_behind = 0;  // 0: synchronous; n >= 2: files in flight
    struct WriteSlot {
        void* pinned = nullptr;
        size_t pinned_bytes = 0;
        std::thread th;
        bool active = false;
        int32_t code = 0;
        std::string msg;
        std::string dest;  // the file the pending write goes to
    };
    std::vector<WriteSlot> write_slots = std::vector<WriteSlot>(1);
    int write_next = 0;
    // legs of the last me_output_mesh(".obj") call, host wall clock: [0] mesh indexing + vertex kernels, [1] text
    // formatting kernels, [2] D2H copy of the text, [3] file write (me_last_mesh_timing)
    double mesh_ms[4] = {0, 0, 0, 0};
    int64_t mesh_bytes = 0;

    // Output overlap (me_ctx_set_output_overlap; BASELINE configs[4]: depth -> stereogram -> mesh per image, many images):
    // the output back end's kernels, copies and their host waits run on `out_stream`, ordered behind the extract_depth step
    // that PRODUCED the depth buffer they read -- not behind whatever the caller has queued on the main stream since -- so
    // image i + 1's depth step runs on the GPU while the host waits for image i's mesh counts, text and D2H copy.  A depth
    // buffer is known by its address range: the step that writes it records `produced` behind itself, every output call
    // on a buffer leaves `consumed` behind itself, and the next step that writes the range waits for that (the caller
    // alternates two buffers).  api.hip OutputScope.
    bool output_overlap = false;
    hipStream_t out_stream = nullptr;
    struct DepthSlot {
        const char* base = nullptr;
        size_t bytes = 0;
        hipEvent_t produced = nullptr, consumed = nullptr;
        bool has_produced = false, has_consumed = false;
        uint64_t stamp = 0;
    };
    DepthSlot depth_slots[4];
    uint64_t depth_stamp = 0;

    // persistent workspaces keyed by site name (no aliasing: zero borders stay zero)
    std::map<std::string, me::DevBuf> bufs;

    // The whole extract_depth step as one hipGraph (shapes are static per batch size).  A call whose pointers
    // all live on the device and that needs no host callback is enqueued eagerly the first time it is seen,
    // captured the second time and replayed with one hipGraphLaunch from then on; anything that changes what the
    // captured launches would do (weights, stream, another batch or another pointer) drops the graph.
    struct GraphKey {
        const void* in = nullptr;
        const void* f_norm = nullptr;
        void* depth = nullptr;
        void* fov = nullptr;
        hipStream_t stream = nullptr;
        int32_t batch = 0, entry = 0;
        uint64_t weights_generation = 0;
        bool operator==(const GraphKey& o) const {
            return in == o.in && f_norm == o.f_norm && depth == o.depth && fov == o.fov && stream == o.stream &&
                   batch == o.batch && entry == o.entry && weights_generation == o.weights_generation;
        }
    };
    bool graph_enabled = false;   // me_ctx_set_graph / ME_GRAPH=1: measured equal (f16) to 1 % slower (fp8) than eager
    bool capturing = false;       // site_buf must not allocate while a capture is open
    bool graph_seen = false, graph_refused = false;
    GraphKey graph_key;           // of graph_exec, or of the eager call last seen
    hipGraphExec_t graph_exec = nullptr;
    uint64_t weights_generation = 0;
    int64_t graph_launches = 0;   // replays so far (me_graph_launch_count)
    void drop_graph() {
        if (graph_exec) {
            // replays are asynchronous: a launch still queued on the stream references the exec's kernarg / node
            // memory, so the stream is drained before the exec goes
            (void)hipStreamSynchronize(stream);
            (void)hipGraphExecDestroy(graph_exec);
        }
        graph_exec = nullptr, graph_seen = false, graph_refused = false;
    }
    // Layout of the weight arena as a hash (FNV-1a over dtype, geometry, split mask and every slot's offset / size):
    // ranks that exchange arenas (me_bcast_weights, me_weights_adopt) must agree on it
    uint64_t arena_layout_hash() const {
        uint64_t h = 1469598103934665603ull;
        auto mix = [&](uint64_t v) {
            for (int i = 0; i < 8; ++i) h = (h ^ ((v >> (8 * i)) & 0xff)) * 1099511628211ull;
        };
        mix((uint64_t)dtype), mix(fp8 ? 1 : 0), mix((uint64_t)split_mask), mix(arena_bytes);
        for (const me::WeightSlot& s : slots) mix(s.offset), mix(s.bytes), mix((uint64_t)s.kind), mix(s.dup ? 1 : 0);
        for (int i = 0; i < 5; ++i) mix(fused_off[i]);
        mix(head_fused_off);
        mix(feat_fused_off);
        return h;
    }

    // geometry helpers
    int g() const { return cfg.grid; }
    int P() const { return cfg.grid * cfg.grid; }
    int T() const { return cfg.grid * cfg.grid + 1; }
    int S() const { return 64 * cfg.grid; }
    int C() const { return cfg.embed_dim; }
};

namespace me {

void build_weight_table(me_ctx* ctx);
void resolve_weights(me_ctx* ctx);
void load_weight(me_ctx* ctx, const char* name, const void* data, int32_t weight_dtype,
                 const int64_t* dims, int32_t ndim);
void finalize_weights(me_ctx* ctx);
void load_checkpoint_pt(me_ctx* ctx, const char* path);
// (re)derives the MX fp8 weight copies from the 16-bit arena (fp8 contexts; no-op otherwise)
void build_fp8_weights(me_ctx* ctx);

// thrown by site_buf when a buffer would have to be (re)allocated while a stream capture is open
struct CaptureAbort {};
// persistent device buffer for a pipeline site; zero-filled when (re)allocated
void* site_buf(me_ctx* ctx, const std::string& name, size_t bytes);

// host-or-device pointer helpers
bool is_device_ptr(const void* p);
// returns a device pointer holding `bytes` of `p` (copying through site buffer `name` if host)
const void* to_device(me_ctx* ctx, const void* p, size_t bytes, const std::string& name);
// device result -> caller pointer (host or device)
void from_device(me_ctx* ctx, void* dst, const void* src_dev, size_t bytes);

// ---- stages (all device pointers; see pipeline.hip) ----
struct VitTaps {
    // called after block `index` with the f32 token stream [W*T][C]
    void (*fn)(void* user, int index, const float* tokens) = nullptr;
    void* user = nullptr;
};
// patches16 [W*P][768] -> tokens32 (residual stream, scratch) ; final LayerNorm written to
// final16 (16-bit) and/or final32.  `tag` names the scratch buffers (one set per stream).
void vit_forward(me_ctx* ctx, int which, const void* patches16, int W, const VitTaps& taps,
                 void* final16, float* final32, const std::string& tag, hipStream_t stream);

// with_fov: the FOV encoder (fov.rs:57-63 ViT + Linear) runs as a third row segment of the encoder's ViT
// launches; stage_fov_tail finishes it
void stage_encoder(me_ctx* ctx, const float* img32, int B, bool with_fov);
// the two halves of stage_encoder: everything up to encodings 2 - 4 (and the FOV encoder's Linear), then the two
// latent upsample chains (encodings 0 and 1, encoder.rs:307-309)
void stage_encoder_trunk(me_ctx* ctx, const float* img32, int B, bool with_fov);
void stage_encoder_latents(me_ctx* ctx, int B);
void stage_decoder(me_ctx* ctx, int B, bool want_features32);
// decoder levels `first` down to `last` (4 = lowest resolution ... 0); stage_decoder = levels 4 to 0
void stage_decoder_levels(me_ctx* ctx, int B, bool want_features32, int first, int last);
void stage_fov_vit(me_ctx* ctx, int B, hipStream_t s);
void stage_fov_tail(me_ctx* ctx, int B, float* fov_deg_dev);
// f_norm_dev [B]; clamp 0 = canonical (no clamp, f_norm ignored -> 1)
void stage_head(me_ctx* ctx, int B, const float* f_norm_dev, bool clamp, float* depth_dev, bool pre_image = false);

void report(me_ctx* ctx, float pos, const char* msg);

// Scope of one output back-end entry point (api.hip): with output overlap on, the context's stream is the output stream
// for the duration of the call, behind the step that produced `depth`; on exit the buffer's `consumed` event is left.
struct OutputScope {
    me_ctx* ctx;
    hipStream_t saved = nullptr;
    int slot = -1;
    bool active = false;
    OutputScope(me_ctx* c, const void* depth);
    ~OutputScope();
    OutputScope(const OutputScope&) = delete;
    OutputScope& operator=(const OutputScope&) = delete;
};

// calibrate.hip: the two fixed loops of bench.py's calibration leg (out[6])
void calibrate(me_ctx* ctx, double* out);

}  // namespace me
