// Checkpoint ingestion: the table of tensors the model expects (PyTorch names and layouts,
// SURVEY App. C / reference mod.rs:185-210 remap rules), packing into one device arena in the
// layouts the kernels read, and the typed views (ModelW) over that arena.
//
// The reference builds each stage's modules, loads and drops them per call (mod.rs:276-351);
// here all 952 M parameters stay resident (1.9 GB of 288 GB) in ONE allocation so that a
// multi-GPU job moves them with a single RCCL broadcast (rccl_bcast.hip).
#include <cstring>

#include "../host/pt_reader.hpp"
#include "model.h"

namespace me {

namespace {

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

void add_slot(me_ctx* ctx, const std::string& name, std::vector<int64_t> dims, PackKind kind,
              bool dup = false) {
    WeightSlot s;
    s.name = name;
    s.dims = std::move(dims);
    s.kind = kind;
    const bool f32 = kind == PK_VEC_F32 || kind == PK_CONVK_F32;
    s.dup = dup && !f32;
    s.bytes = (size_t)s.numel() * (f32 ? 4 : 2) * (s.dup ? 2 : 1);
    s.offset = ctx->arena_bytes;
    ctx->arena_bytes = align_up(ctx->arena_bytes + s.bytes, 256);
    ctx->slot_by_name[name] = (int)ctx->slots.size();
    ctx->slots.push_back(std::move(s));
}

void add_vit(me_ctx* ctx, const std::string& p) {
    const int64_t C = ctx->C(), T = ctx->T();
    add_slot(ctx, p + "cls_token", {1, 1, C}, PK_VEC_F32);
    add_slot(ctx, p + "pos_embed", {1, T, C}, PK_VEC_F32);
    add_slot(ctx, p + "patch_embed.proj.weight", {C, 3, 16, 16}, PK_MAT_16);
    add_slot(ctx, p + "patch_embed.proj.bias", {C}, PK_VEC_F32);
    for (int i = 0; i < ctx->cfg.depth; ++i) {
        const std::string b = p + "blocks." + std::to_string(i) + ".";
        add_slot(ctx, b + "norm1.weight", {C}, PK_VEC_F32);
        add_slot(ctx, b + "norm1.bias", {C}, PK_VEC_F32);
        add_slot(ctx, b + "attn.qkv.weight", {3 * C, C}, PK_MAT_16);
        add_slot(ctx, b + "attn.qkv.bias", {3 * C}, PK_VEC_F32);
        add_slot(ctx, b + "attn.proj.weight", {C, C}, PK_MAT_16);
        add_slot(ctx, b + "attn.proj.bias", {C}, PK_VEC_F32);
        add_slot(ctx, b + "ls1.gamma", {C}, PK_VEC_F32);
        add_slot(ctx, b + "norm2.weight", {C}, PK_VEC_F32);
        add_slot(ctx, b + "norm2.bias", {C}, PK_VEC_F32);
        add_slot(ctx, b + "mlp.fc1.weight", {4 * C, C}, PK_MAT_16);
        add_slot(ctx, b + "mlp.fc1.bias", {4 * C}, PK_VEC_F32);
        add_slot(ctx, b + "mlp.fc2.weight", {C, 4 * C}, PK_MAT_16);
        add_slot(ctx, b + "mlp.fc2.bias", {C}, PK_VEC_F32);
        add_slot(ctx, b + "ls2.gamma", {C}, PK_VEC_F32);
    }
    add_slot(ctx, p + "norm.weight", {C}, PK_VEC_F32);
    add_slot(ctx, p + "norm.bias", {C}, PK_VEC_F32);
}

// encoder.rs:85-118 init_project_upsample_block
void add_upsample(me_ctx* ctx, const std::string& p, int64_t dim_out, int layers, int64_t dim_int) {
    const bool dup = ctx->split(SPLIT_UPSAMPLE);
    add_slot(ctx, p + "0.weight", {dim_int, ctx->C(), 1, 1}, PK_MAT_16, dup);
    for (int i = 0; i < layers; ++i) {
        const int64_t in = i == 0 ? dim_int : dim_out;
        add_slot(ctx, p + std::to_string(i + 1) + ".weight", {in, dim_out, 2, 2}, PK_CONVT_16, dup);
    }
}

const float* fptr(me_ctx* ctx, const std::string& name) {
    auto it = ctx->slot_by_name.find(name);
    ME_CHECK(it != ctx->slot_by_name.end(), ME_ERR_BAD_ARG, "internal: no slot %s", name.c_str());
    return reinterpret_cast<const float*>(ctx->arena + ctx->slots[it->second].offset);
}
const void* vptr(me_ctx* ctx, const std::string& name) { return fptr(ctx, name); }

void resolve_vit(me_ctx* ctx, const std::string& p, VitW& v) {
    v.patch_w = vptr(ctx, p + "patch_embed.proj.weight");
    v.patch_b = fptr(ctx, p + "patch_embed.proj.bias");
    v.cls = fptr(ctx, p + "cls_token");
    v.pos = fptr(ctx, p + "pos_embed");
    v.norm_w = fptr(ctx, p + "norm.weight");
    v.norm_b = fptr(ctx, p + "norm.bias");
    v.blocks.resize(ctx->cfg.depth);
    for (int i = 0; i < ctx->cfg.depth; ++i) {
        const std::string b = p + "blocks." + std::to_string(i) + ".";
        VitBlockW& w = v.blocks[i];
        w.ln1_w = fptr(ctx, b + "norm1.weight"), w.ln1_b = fptr(ctx, b + "norm1.bias");
        w.qkv_w = vptr(ctx, b + "attn.qkv.weight"), w.qkv_b = fptr(ctx, b + "attn.qkv.bias");
        w.proj_w = vptr(ctx, b + "attn.proj.weight"), w.proj_b = fptr(ctx, b + "attn.proj.bias");
        w.ls1 = fptr(ctx, b + "ls1.gamma");
        w.ln2_w = fptr(ctx, b + "norm2.weight"), w.ln2_b = fptr(ctx, b + "norm2.bias");
        w.fc1_w = vptr(ctx, b + "mlp.fc1.weight"), w.fc1_b = fptr(ctx, b + "mlp.fc1.bias");
        w.fc2_w = vptr(ctx, b + "mlp.fc2.weight"), w.fc2_b = fptr(ctx, b + "mlp.fc2.bias");
        w.ls2 = fptr(ctx, b + "ls2.gamma");
    }
}

void resolve_upsample(me_ctx* ctx, const std::string& p, int64_t dim_out, int layers,
                      int64_t dim_int, UpsampleW& u) {
    u.conv = vptr(ctx, p + "0.weight");
    u.dim_int = (int)dim_int;
    u.convt.clear(), u.cin.clear(), u.cout.clear();
    for (int i = 0; i < layers; ++i) {
        u.convt.push_back(vptr(ctx, p + std::to_string(i + 1) + ".weight"));
        u.cin.push_back((int)(i == 0 ? dim_int : dim_out));
        u.cout.push_back((int)dim_out);
    }
}

// IEEE half <-> float on the host (no dependence on compiler half support in host code paths)
inline float half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000) << 16;
    uint32_t exp = (h >> 10) & 0x1f, man = h & 0x3ff;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else {
            int e = -1;
            do {
                ++e;
                man <<= 1;
            } while (!(man & 0x400));
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ff) << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | (man << 13);
    } else {
        bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}
inline uint16_t float_to_half(float f) {
    _Float16 h = (_Float16)f;  // round to nearest even
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}
inline uint16_t float_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1)) >> 16);
}

struct HostSrc {
    const void* p;
    int32_t dt;
    float get(int64_t i) const {
        switch (dt) {
            case ME_WEIGHT_F32: return ((const float*)p)[i];
            case ME_WEIGHT_F16: return half_to_float(((const uint16_t*)p)[i]);
            case ME_WEIGHT_BF16: {
                const uint32_t u = (uint32_t)((const uint16_t*)p)[i] << 16;
                float f;
                memcpy(&f, &u, 4);
                return f;
            }
            default: return (float)((const double*)p)[i];
        }
    }
};
size_t weight_elem_size(int32_t dt) {
    return dt == ME_WEIGHT_F32 ? 4 : (dt == ME_WEIGHT_F64 ? 8 : 2);
}

}  // namespace

void build_weight_table(me_ctx* ctx) {
    const me_model_config& c = ctx->cfg;
    const int64_t C = c.embed_dim, dec = c.dec_dim;
    const int64_t e0 = c.enc_dims[0], e1 = c.enc_dims[1], e2 = c.enc_dims[2], e3 = c.enc_dims[3];
    ctx->slots.clear();
    ctx->slot_by_name.clear();
    ctx->arena_bytes = 0;
    add_vit(ctx, "encoder.patch_encoder.");
    add_vit(ctx, "encoder.image_encoder.");
    // encoder.rs:48-71
    add_upsample(ctx, "encoder.upsample_latent0.", dec, 3, e0);
    add_upsample(ctx, "encoder.upsample_latent1.", e0, 2, e0);
    add_upsample(ctx, "encoder.upsample0.", e1, 1, e1);
    add_upsample(ctx, "encoder.upsample1.", e2, 1, e2);
    add_upsample(ctx, "encoder.upsample2.", e3, 1, e3);
    add_slot(ctx, "encoder.upsample_lowres.weight", {C, e3, 2, 2}, PK_CONVT_16, ctx->split(SPLIT_UPSAMPLE));
    add_slot(ctx, "encoder.upsample_lowres.bias", {e3}, PK_VEC_F32);
    add_slot(ctx, "encoder.fuse_lowres.weight", {e3, 2 * e3, 1, 1}, PK_MAT_16, ctx->split(SPLIT_UPSAMPLE));
    add_slot(ctx, "encoder.fuse_lowres.bias", {e3}, PK_VEC_F32);
    // decoder.rs:115-146; dims_encoder = [dec, e0, e1, e2, e3] (mod.rs:293-295).  PyTorch keeps an
    // Identity at convs.0 when dims_encoder[0] == dim_decoder, so the file's keys start at convs.1
    const int64_t dims_enc[5] = {dec, e0, e1, e2, e3};
    ME_CHECK(dims_enc[0] == dec, ME_ERR_BAD_SHAPE, "decoder: dims_encoder[0] != dim_decoder");
    for (int i = 1; i < 5; ++i)
        add_slot(ctx, "decoder.convs." + std::to_string(i) + ".weight", {dec, dims_enc[i], 3, 3},
                 PK_CONV_16, ctx->split(SPLIT_DEC_CONVS));
    for (int i = 0; i < 5; ++i) {
        const std::string f = "decoder.fusions." + std::to_string(i) + ".";
        for (const char* rn : {"resnet1", "resnet2"})
            for (const char* idx : {"1", "3"}) {
                add_slot(ctx, f + rn + ".residual." + idx + ".weight", {dec, dec, 3, 3}, PK_CONV_16);
                add_slot(ctx, f + rn + ".residual." + idx + ".bias", {dec}, PK_VEC_F32);
            }
        if (i != 0) add_slot(ctx, f + "deconv.weight", {dec, dec, 2, 2}, PK_CONVT_16, ctx->split(SPLIT_FUSION_OUT));
        add_slot(ctx, f + "out_conv.weight", {dec, dec, 1, 1}, PK_MAT_16, ctx->split(SPLIT_FUSION_OUT));
        add_slot(ctx, f + "out_conv.bias", {dec}, PK_VEC_F32);
    }
    // mod.rs:57-97 (PyTorch Sequential indices 0,1,2,4: index 3 is the ReLU)
    const bool dup_head = ctx->split(SPLIT_HEAD);
    add_slot(ctx, "head.0.weight", {dec / 2, dec, 3, 3}, PK_CONV_16, dup_head);
    add_slot(ctx, "head.0.bias", {dec / 2}, PK_VEC_F32);
    add_slot(ctx, "head.1.weight", {dec / 2, dec / 2, 2, 2}, PK_CONVT_16, dup_head);
    add_slot(ctx, "head.1.bias", {dec / 2}, PK_VEC_F32);
    add_slot(ctx, "head.2.weight", {c.head_dims[0], dec / 2, 3, 3}, PK_CONV_16, dup_head);
    add_slot(ctx, "head.2.bias", {c.head_dims[0]}, PK_VEC_F32);
    add_slot(ctx, "head.4.weight", {c.head_dims[1], c.head_dims[0], 1, 1}, PK_VEC_F32);
    add_slot(ctx, "head.4.bias", {c.head_dims[1]}, PK_VEC_F32);
    // fov.rs:95-128
    add_vit(ctx, "fov.encoder.0.");
    add_slot(ctx, "fov.encoder.1.weight", {dec / 2, C}, PK_MAT_16);
    add_slot(ctx, "fov.encoder.1.bias", {dec / 2}, PK_VEC_F32);
    add_slot(ctx, "fov.downsample.0.weight", {dec / 2, dec, 3, 3}, PK_CONV_16);
    add_slot(ctx, "fov.downsample.0.bias", {dec / 2}, PK_VEC_F32);
    add_slot(ctx, "fov.head.0.weight", {dec / 4, dec / 2, 3, 3}, PK_CONV_16);
    add_slot(ctx, "fov.head.0.bias", {dec / 4}, PK_VEC_F32);
    add_slot(ctx, "fov.head.2.weight", {dec / 8, dec / 4, 3, 3}, PK_CONV_16);
    add_slot(ctx, "fov.head.2.bias", {dec / 8}, PK_VEC_F32);
    const int64_t k = c.grid / 4;  // 6 for grid 24 (fov.rs:115)
    add_slot(ctx, "fov.head.4.weight", {1, dec / 8, k, k}, PK_CONVK_F32);
    add_slot(ctx, "fov.head.4.bias", {1}, PK_VEC_F32);
    // derived: out_conv o deconv of fusion levels 1-4 as one ConvTranspose, [4 dec][3 dec] 16-bit each (part of
    // the arena, so a broadcast carries it; not a checkpoint tensor, so not in the slot table)
    for (int i = 0; i < 5; ++i) ctx->fused_off[i] = 0;
    if (ctx->split(SPLIT_FUSION_OUT))
        for (int i = 1; i < 5; ++i) {
            ctx->fused_off[i] = ctx->arena_bytes;
            ctx->arena_bytes = align_up(ctx->arena_bytes + (size_t)(4 * dec) * (3 * dec) * 2, 256);
        }
    // derived: head.1 o head.2 as one 3x3 convolution with 4 x 32 output channels (compose_head), + its f32 bias tables
    ctx->head_fused_off = 0;
    if (!ctx->split(SPLIT_HEAD)) {
        ctx->head_fused_off = ctx->arena_bytes;
        ctx->arena_bytes = align_up(ctx->arena_bytes + (size_t)128 * 9 * (dec / 2) * 2 + (size_t)(32 + 9 * 32) * 4, 256);
    }
    // derived: fusions.0.out_conv o head.0 as one 3x3 convolution (compose_features), + its f32 bias tables [dec / 2] + [9][dec / 2]
    ctx->feat_fused_off = 0;
    if (!ctx->split(SPLIT_HEAD)) {
        ctx->feat_fused_off = ctx->arena_bytes;
        ctx->arena_bytes = align_up(ctx->arena_bytes + (size_t)(dec / 2) * 9 * dec * 2 + (size_t)(10 * (dec / 2)) * 4, 256);
    }
}

void resolve_weights(me_ctx* ctx) {
    const me_model_config& c = ctx->cfg;
    ModelW& w = ctx->w;
    resolve_vit(ctx, "encoder.patch_encoder.", w.vit[ME_VIT_PATCH_ENCODER]);
    resolve_vit(ctx, "encoder.image_encoder.", w.vit[ME_VIT_IMAGE_ENCODER]);
    resolve_vit(ctx, "fov.encoder.0.", w.vit[ME_VIT_FOV_ENCODER]);
    resolve_upsample(ctx, "encoder.upsample_latent0.", c.dec_dim, 3, c.enc_dims[0], w.up_latent0);
    resolve_upsample(ctx, "encoder.upsample_latent1.", c.enc_dims[0], 2, c.enc_dims[0], w.up_latent1);
    resolve_upsample(ctx, "encoder.upsample0.", c.enc_dims[1], 1, c.enc_dims[1], w.up0);
    resolve_upsample(ctx, "encoder.upsample1.", c.enc_dims[2], 1, c.enc_dims[2], w.up1);
    resolve_upsample(ctx, "encoder.upsample2.", c.enc_dims[3], 1, c.enc_dims[3], w.up2);
    w.up_lowres_w = vptr(ctx, "encoder.upsample_lowres.weight");
    w.up_lowres_b = fptr(ctx, "encoder.upsample_lowres.bias");
    w.fuse_w = vptr(ctx, "encoder.fuse_lowres.weight");
    w.fuse_b = fptr(ctx, "encoder.fuse_lowres.bias");
    w.dec_convs[0] = nullptr;
    for (int i = 1; i < 5; ++i)
        w.dec_convs[i] = vptr(ctx, "decoder.convs." + std::to_string(i) + ".weight");
    for (int i = 0; i < 5; ++i) {
        const std::string f = "decoder.fusions." + std::to_string(i) + ".";
        FusionW& fw = w.fusions[i];
        RcuW* rc[2] = {&fw.resnet1, &fw.resnet2};
        const char* rn[2] = {"resnet1", "resnet2"};
        const char* idx[2] = {"1", "3"};
        for (int r = 0; r < 2; ++r)
            for (int k = 0; k < 2; ++k) {
                rc[r]->w[k] = vptr(ctx, f + rn[r] + ".residual." + idx[k] + ".weight");
                rc[r]->b[k] = fptr(ctx, f + rn[r] + ".residual." + idx[k] + ".bias");
            }
        fw.deconv = i != 0 ? vptr(ctx, f + "deconv.weight") : nullptr;
        fw.fused_w = ctx->fused_off[i] ? (const void*)(ctx->arena + ctx->fused_off[i]) : nullptr;
        fw.out_w = vptr(ctx, f + "out_conv.weight");
        fw.out_b = fptr(ctx, f + "out_conv.bias");
    }
    w.head0_w = vptr(ctx, "head.0.weight"), w.head0_b = fptr(ctx, "head.0.bias");
    w.head1_w = vptr(ctx, "head.1.weight"), w.head1_b = fptr(ctx, "head.1.bias");
    w.head2_w = vptr(ctx, "head.2.weight"), w.head2_b = fptr(ctx, "head.2.bias");
    w.head4_w = fptr(ctx, "head.4.weight"), w.head4_b = fptr(ctx, "head.4.bias");
    w.head_fused_w = ctx->head_fused_off ? (const void*)(ctx->arena + ctx->head_fused_off) : nullptr;
    w.head_fused_b = ctx->head_fused_off ? (const float*)(ctx->arena + ctx->head_fused_off + (size_t)128 * 9 * (c.dec_dim / 2) * 2) : nullptr;
    w.feat_fused_w = ctx->feat_fused_off ? (const void*)(ctx->arena + ctx->feat_fused_off) : nullptr;
    w.feat_fused_b = ctx->feat_fused_off ? (const float*)(ctx->arena + ctx->feat_fused_off + (size_t)(c.dec_dim / 2) * 9 * c.dec_dim * 2) : nullptr;
    w.fov_lin_w = vptr(ctx, "fov.encoder.1.weight"), w.fov_lin_b = fptr(ctx, "fov.encoder.1.bias");
    w.fov_down_w = vptr(ctx, "fov.downsample.0.weight");
    w.fov_down_b = fptr(ctx, "fov.downsample.0.bias");
    w.fov_h0_w = vptr(ctx, "fov.head.0.weight"), w.fov_h0_b = fptr(ctx, "fov.head.0.bias");
    w.fov_h2_w = vptr(ctx, "fov.head.2.weight"), w.fov_h2_b = fptr(ctx, "fov.head.2.bias");
    w.fov_h4_w = fptr(ctx, "fov.head.4.weight"), w.fov_h4_b = fptr(ctx, "fov.head.4.bias");
}

void load_weight(me_ctx* ctx, const char* name, const void* data, int32_t weight_dtype,
                 const int64_t* dims, int32_t ndim) {
    ME_CHECK(name && data && dims, ME_ERR_BAD_ARG, "me_load_weight: null argument");
    ME_CHECK(weight_dtype >= ME_WEIGHT_F32 && weight_dtype <= ME_WEIGHT_F64, ME_ERR_BAD_ARG,
             "me_load_weight: bad weight dtype %d", weight_dtype);
    auto it = ctx->slot_by_name.find(name);
    ME_CHECK(it != ctx->slot_by_name.end(), ME_ERR_BAD_WEIGHT, "unexpected tensor '%s'", name);
    WeightSlot& s = ctx->slots[it->second];
    bool same = (int)s.dims.size() == ndim;
    for (int i = 0; same && i < ndim; ++i) same = s.dims[i] == dims[i];
    if (!same) {
        std::string got, want;
        for (int i = 0; i < ndim; ++i) got += (i ? "," : "") + std::to_string(dims[i]);
        for (size_t i = 0; i < s.dims.size(); ++i) want += (i ? "," : "") + std::to_string(s.dims[i]);
        fail(ME_ERR_BAD_WEIGHT, "tensor '%s' has shape [%s], expected [%s]", name, got.c_str(),
             want.c_str());
    }
    const int64_t n = s.numel();
    const size_t src_bytes = (size_t)n * weight_elem_size(weight_dtype);
    std::vector<char> staged;
    if (is_device_ptr(data)) {
        staged.resize(src_bytes);
        ME_HIP(hipMemcpy(staged.data(), data, src_bytes, hipMemcpyDeviceToHost));
        data = staged.data();
    }
    const HostSrc src{data, weight_dtype};
    if (ctx->split(SPLIT_FUSION_OUT) && s.name.rfind("decoder.fusions.", 0) == 0 &&
        (s.name.find(".deconv.weight") != std::string::npos || s.name.find(".out_conv.weight") != std::string::npos)) {
        std::vector<float>& keep = ctx->factor_keep[s.name];  // composed at finalize (compose_fusion_out)
        keep.resize((size_t)n);
        for (int64_t i = 0; i < n; ++i) keep[i] = src.get(i);
    }
    if (ctx->head_fused_off && (s.name == "head.1.weight" || s.name == "head.1.bias" || s.name == "head.2.weight" || s.name == "head.2.bias")) {
        std::vector<float>& keep = ctx->factor_keep[s.name];  // composed at finalize (compose_head)
        keep.resize((size_t)n);
        for (int64_t i = 0; i < n; ++i) keep[i] = src.get(i);
    }
    if (ctx->feat_fused_off && (s.name == "decoder.fusions.0.out_conv.weight" || s.name == "decoder.fusions.0.out_conv.bias" ||
                                s.name == "head.0.weight" || s.name == "head.0.bias")) {
        std::vector<float>& keep = ctx->factor_keep[s.name];  // composed at finalize (compose_features)
        keep.resize((size_t)n);
        for (int64_t i = 0; i < n; ++i) keep[i] = src.get(i);
    }
    std::vector<char> packed(s.bytes);
    const bool to_bf16 = ctx->dtype == ME_DTYPE_BF16;
    auto put16 = [&](int64_t dst, int64_t si) {
        uint16_t* o = reinterpret_cast<uint16_t*>(packed.data());
        if (!to_bf16 && weight_dtype == ME_WEIGHT_F16)
            o[dst] = ((const uint16_t*)data)[si];  // fp16 checkpoint, f16 operands: exact
        else
            o[dst] = to_bf16 ? float_to_bf16(src.get(si)) : float_to_half(src.get(si));
    };
    switch (s.kind) {
        case PK_VEC_F32: {
            float* o = reinterpret_cast<float*>(packed.data());
            if (weight_dtype == ME_WEIGHT_F32)
                memcpy(o, data, (size_t)n * 4);
            else
                for (int64_t i = 0; i < n; ++i) o[i] = src.get(i);
            break;
        }
        // dup: every run of K (or Cin) source values is followed by a copy of itself: [W | W]
        case PK_MAT_16:
            if (s.dup) {
                const int64_t N = s.dims[0], K = n / N;
                for (int64_t r = 0; r < N; ++r)
                    for (int64_t k = 0; k < K; ++k) put16(r * 2 * K + k, r * K + k), put16(r * 2 * K + K + k, r * K + k);
            } else if (!to_bf16 && weight_dtype == ME_WEIGHT_F16) {
                memcpy(packed.data(), data, (size_t)n * 2);
            } else {
                for (int64_t i = 0; i < n; ++i) put16(i, i);
            }
            break;
        case PK_CONV_16: {  // [Cout][Cin][kh][kw] -> [Cout][kh*kw][Cin]
            const int64_t Cout = s.dims[0], Cin = s.dims[1], kk = s.dims[2] * s.dims[3];
            const int64_t D = s.dup ? 2 : 1;
            for (int64_t co = 0; co < Cout; ++co)
                for (int64_t ci = 0; ci < Cin; ++ci)
                    for (int64_t t = 0; t < kk; ++t)
                        for (int64_t d = 0; d < D; ++d)
                            put16((co * kk + t) * D * Cin + d * Cin + ci, (co * Cin + ci) * kk + t);
            break;
        }
        case PK_CONVT_16: {  // [Cin][Cout][2][2] -> [(dy*2+dx)*Cout + co][Cin]
            const int64_t Cin = s.dims[0], Cout = s.dims[1];
            const int64_t D = s.dup ? 2 : 1;
            for (int64_t ci = 0; ci < Cin; ++ci)
                for (int64_t co = 0; co < Cout; ++co)
                    for (int64_t q = 0; q < 4; ++q)
                        for (int64_t d = 0; d < D; ++d)
                            put16((q * Cout + co) * D * Cin + d * Cin + ci, (ci * Cout + co) * 4 + q);
            break;
        }
        case PK_CONVK_F32: {  // [1][Cin][k][k] -> f32 [k][k][Cin]
            const int64_t Cin = s.dims[1], kk = s.dims[2] * s.dims[3];
            float* o = reinterpret_cast<float*>(packed.data());
            for (int64_t ci = 0; ci < Cin; ++ci)
                for (int64_t t = 0; t < kk; ++t) o[t * Cin + ci] = src.get(ci * kk + t);
            break;
        }
    }
    ME_HIP(hipMemcpy(ctx->arena + s.offset, packed.data(), s.bytes, hipMemcpyHostToDevice));
    s.loaded = true;
    ctx->finalized = false;
}

// mod.rs:229-249 load_record: PytorchStore::from_file + apply + the errors / missing checks
void load_checkpoint_pt(me_ctx* ctx, const char* path) {
    ME_CHECK(path, ME_ERR_BAD_ARG, "me_load_checkpoint_pt: null path");
    ctx->unused_weights.clear();
    try {
        matrix_eyes::PtFile file(path);
        for (const matrix_eyes::PtTensor& t : file.tensors()) {
            // a key the model has no slot for is not an error (mod.rs:236-243 checks errors and missing only):
            // state_dict wrapper siblings, EMA copies, integer buffers -- whatever their dtype
            if (ctx->slot_by_name.find(t.name) == ctx->slot_by_name.end()) {
                ctx->unused_weights.push_back(t.name);
                continue;
            }
            int32_t wd;
            if (t.dtype == "f16") wd = ME_WEIGHT_F16;
            else if (t.dtype == "f32") wd = ME_WEIGHT_F32;
            else if (t.dtype == "bf16") wd = ME_WEIGHT_BF16;
            else if (t.dtype == "f64") wd = ME_WEIGHT_F64;
            else fail(ME_ERR_BAD_WEIGHT, "%s: tensor %s has dtype %s", path, t.name.c_str(), t.dtype.c_str());
            load_weight(ctx, t.name.c_str(), t.data, wd, t.dims.data(), (int32_t)t.dims.size());
        }
    } catch (const matrix_eyes::CheckpointError& err) {  // LoaderError::Pytorch
        fail(ME_ERR_IO, "failed to load checkpoint: %s", err.what());
    }
    finalize_weights(ctx);
}

// A factor of a composed layer in the checkpoint's layout: the host copy load_weight kept, or -- when the arena came by
// me_weights_adopt / me_bcast_weights and only some other factor was reloaded -- read back from its packed arena slot (the
// 16-bit value the un-composed path would use), so that composed weights never lag behind a factor loaded later.
static const std::vector<float>& fetch_factor(me_ctx* ctx, const std::string& name) {
    auto it = ctx->factor_keep.find(name);
    if (it != ctx->factor_keep.end()) return it->second;
    const bool to_bf16 = ctx->dtype == ME_DTYPE_BF16;
    auto from16 = [&](uint16_t h) {
        if (!to_bf16) return half_to_float(h);
        const uint32_t u = (uint32_t)h << 16;
        float f;
        memcpy(&f, &u, 4);
        return f;
    };
    const WeightSlot& s = ctx->slots[ctx->slot_by_name.at(name)];
    std::vector<float> out((size_t)s.numel());
    if (s.kind == PK_VEC_F32) {
        ME_HIP(hipMemcpy(out.data(), ctx->arena + s.offset, s.bytes, hipMemcpyDeviceToHost));
    } else {
        std::vector<uint16_t> raw(s.bytes / 2);
        ME_HIP(hipMemcpy(raw.data(), ctx->arena + s.offset, s.bytes, hipMemcpyDeviceToHost));
        const int64_t Dd = s.dup ? 2 : 1;
        if (s.kind == PK_CONVT_16) {  // [(q*Cout + co)][Dd * Cin] -> [Cin][Cout][q]
            const int64_t Cin = s.dims[0], Cout = s.dims[1];
            for (int64_t ci = 0; ci < Cin; ++ci)
                for (int64_t co = 0; co < Cout; ++co)
                    for (int64_t q = 0; q < 4; ++q) out[(ci * Cout + co) * 4 + q] = from16(raw[(q * Cout + co) * Dd * Cin + ci]);
        } else if (s.kind == PK_CONV_16) {  // [Cout][kk][Dd * Cin] -> [Cout][Cin][kk]
            const int64_t Cout = s.dims[0], Cin = s.dims[1], kk = s.dims[2] * s.dims[3];
            for (int64_t co = 0; co < Cout; ++co)
                for (int64_t ci = 0; ci < Cin; ++ci)
                    for (int64_t t = 0; t < kk; ++t) out[(co * Cin + ci) * kk + t] = from16(raw[(co * kk + t) * Dd * Cin + ci]);
        } else {  // PK_MAT_16 [N][Dd * K] -> [N][K]
            const int64_t N = s.dims[0], K = s.numel() / N;
            for (int64_t r = 0; r < N; ++r)
                for (int64_t k = 0; k < K; ++k) out[r * K + k] = from16(raw[r * Dd * K + k]);
        }
    }
    return ctx->factor_keep.emplace(name, std::move(out)).first->second;
}

// decoder.rs:95-101: `deconv` (ConvTranspose 2x2 stride 2, no bias) then `out_conv` (1x1 + bias) with nothing
// between them -- one linear map per output position q = (dy, dx): W'_q[co2][ci] = sum_co W_out[co2][co] *
// W_deconv[ci][co][q].  Composed in f64 from the checkpoint's values and stored as a 16-bit hi + lo pair, so the
// composed weights are as exact as the activations beside them ([hi | lo | hi] against [W_hi | W_hi | W_lo]).
// It removes the 4x-resolution intermediate of every fusion level (604 MB written and read back at level 1) and
// one launch per level.
static void compose_fusion_out(me_ctx* ctx) {
    if (!ctx->split(SPLIT_FUSION_OUT)) return;
    const int64_t D = ctx->cfg.dec_dim;
    const bool to_bf16 = ctx->dtype == ME_DTYPE_BF16;
    auto to16 = [&](float f) { return to_bf16 ? float_to_bf16(f) : float_to_half(f); };
    auto from16 = [&](uint16_t h) {
        if (!to_bf16) return half_to_float(h);
        const uint32_t u = (uint32_t)h << 16;
        float f;
        memcpy(&f, &u, 4);
        return f;
    };
    for (int i = 1; i < 5; ++i) {
        const std::string f = "decoder.fusions." + std::to_string(i) + ".";
        const std::string dn = f + "deconv.weight", on = f + "out_conv.weight";
        auto d = ctx->factor_keep.find(dn);
        auto o = ctx->factor_keep.find(on);
        if (d == ctx->factor_keep.end() && o == ctx->factor_keep.end()) continue;  // nothing reloaded: arena's own
        if (!ctx->slots[ctx->slot_by_name.at(dn)].loaded || !ctx->slots[ctx->slot_by_name.at(on)].loaded)
            continue;  // finalize reports the missing one
        const std::vector<float>&Wd = fetch_factor(ctx, dn), &Wo = fetch_factor(ctx, on);  // [ci][co][q], [co2][co]
        std::vector<uint16_t> packed((size_t)(4 * D) * (3 * D));
        std::vector<double> wq((size_t)D * D), acc((size_t)D);
        for (int64_t q = 0; q < 4; ++q) {
            for (int64_t co = 0; co < D; ++co)  // W_deconv for this position as [co][ci]: the inner loops run along ci
                for (int64_t ci = 0; ci < D; ++ci) wq[co * D + ci] = (double)Wd[(ci * D + co) * 4 + q];
            for (int64_t co2 = 0; co2 < D; ++co2) {
                std::fill(acc.begin(), acc.end(), 0.0);
                for (int64_t co = 0; co < D; ++co) {
                    const double w = (double)Wo[co2 * D + co];
                    const double* row = &wq[co * D];
                    for (int64_t ci = 0; ci < D; ++ci) acc[ci] += w * row[ci];
                }
                uint16_t* dst = &packed[(size_t)(q * D + co2) * (3 * D)];
                for (int64_t ci = 0; ci < D; ++ci) {
                    const uint16_t hi = to16((float)acc[ci]);
                    const uint16_t lo = to16((float)(acc[ci] - (double)from16(hi)));
                    dst[ci] = hi, dst[D + ci] = hi, dst[2 * D + ci] = lo;
                }
            }
        }
        ME_HIP(hipMemcpy(ctx->arena + ctx->fused_off[i], packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
    }
    // the factors stay (2 MB of host memory): a later me_load_weight of one of them recomposes at the next finalize
}

// mod.rs:73-94,326-333: head[1] = ConvTranspose2d(128 -> 128, 2 x 2, stride 2, bias) and head[2] = Conv2d(128 -> 32, 3 x 3, padding 1,
// bias) have nothing between them, so they are ONE linear map from the half-resolution map X to the full-resolution one.  An
// output pixel (2y + dy, 2x + dx) reads the ConvTranspose output U at rows 2y + dy + ky - 1 = 2 (y + iy) + ry, i.e. input rows
// y + iy with iy in {-1, 0, 1} -- each output PHASE (dy, dx) is a convolution of X over a 2 x 2 window inside the 3 x 3 one:
//   V[2y + dy][2x + dx][co] = b'[co] + sum_{iy, ix, ci} X[y + iy][x + ix][ci] * W'[(dy, dx), co][iy + 1][ix + 1][ci],
//   W'[(dy, dx), co][iy + 1][ix + 1][ci] = sum over the (ky, kx) that land on (iy, ry), (ix, rx) of
//                                          sum_c Wt[ci][c][ry][rx] * W3[co][c][ky][kx],
// a 3 x 3 convolution with 4 x 32 output channels on the 768 x 768 map (the halo tile with 128 channels) -- the 605 MB
// [1, 128, 1536, 1536] tensor of mod.rs:328 is never written or read (SURVEY 7 "hard part"), and the
// ConvTranspose's launch and FLOPs go.  Bias: b'[co] = b3[co] + sum_{ky, kx} Cb[ky][kx][co], Cb = sum_c bT[c] W3[co][c][ky][kx];
// where tap (ky, kx) of the 3 x 3 convolution falls outside the full-resolution image (its zero padding) the epilogue takes
// that tap's share out again (X is zero-bordered, so the X part vanishes by itself).  Composed in f64 from the checkpoint's
// values, rounded once to the operand type; rows of output channels beyond head_dims[0] are zero.
static void compose_head(me_ctx* ctx) {
    if (!ctx->head_fused_off) return;
    const int64_t Cm = ctx->cfg.dec_dim / 2, Co = ctx->cfg.head_dims[0];
    const bool to_bf16 = ctx->dtype == ME_DTYPE_BF16;
    auto to16 = [&](float f) { return to_bf16 ? float_to_bf16(f) : float_to_half(f); };
    const char* names[4] = {"head.1.weight", "head.1.bias", "head.2.weight", "head.2.bias"};
    bool any = false;
    for (const char* nm : names) any = any || ctx->factor_keep.count(nm);
    if (!any) return;  // nothing (re)loaded on this context: the arena's own composition (me_weights_adopt / broadcast)
    for (const char* nm : names)
        if (!ctx->slots[ctx->slot_by_name.at(nm)].loaded) return;  // finalize reports the missing one
    const std::vector<float>& Wt = fetch_factor(ctx, "head.1.weight");  // [ci][c][dy][dx]
    const std::vector<float>& bT = fetch_factor(ctx, "head.1.bias");    // [c]
    const std::vector<float>& W3 = fetch_factor(ctx, "head.2.weight");  // [co][c][ky][kx]
    const std::vector<float>& b3 = fetch_factor(ctx, "head.2.bias");    // [co]
    std::vector<double> Wp((size_t)128 * 9 * Cm, 0.0);      // [(phase * 32 + co)][tap][ci]
    std::vector<double> w3row((size_t)Cm);
    for (int dy = 0; dy < 2; ++dy)
        for (int dx = 0; dx < 2; ++dx)
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) {
                    const int ay = dy + ky - 1, ax = dx + kx - 1;
                    const int iy = ay < 0 ? -1 : ay / 2, ix = ax < 0 ? -1 : ax / 2;  // floor(a / 2) for a in -1 .. 2
                    const int ry = ay - 2 * iy, rx = ax - 2 * ix;
                    const int tap = (iy + 1) * 3 + (ix + 1), q = ry * 2 + rx;
                    for (int64_t co = 0; co < Co; ++co) {
                        for (int64_t c = 0; c < Cm; ++c) w3row[c] = (double)W3[((co * Cm + c) * 3 + ky) * 3 + kx];
                        double* dst = &Wp[(((size_t)(dy * 2 + dx) * 32 + co) * 9 + tap) * Cm];
                        for (int64_t ci = 0; ci < Cm; ++ci) {
                            const float* wt = &Wt[(size_t)ci * Cm * 4 + q];  // Wt[ci][c][ry][rx], c strided by 4
                            double a = 0.0;
                            for (int64_t c = 0; c < Cm; ++c) a += (double)wt[c * 4] * w3row[c];
                            dst[ci] += a;
                        }
                    }
                }
    std::vector<uint16_t> packed((size_t)128 * 9 * Cm);
    for (size_t i = 0; i < packed.size(); ++i) packed[i] = to16((float)Wp[i]);
    std::vector<float> tab((size_t)32 + 9 * 32, 0.f);  // [32] interior bias, [9][32] per-tap shares
    for (int64_t co = 0; co < Co; ++co) {
        double bsum = (double)b3[co];
        for (int t = 0; t < 9; ++t) {
            double a = 0.0;
            for (int64_t c = 0; c < Cm; ++c) a += (double)bT[c] * (double)W3[(co * Cm + c) * 9 + t];
            tab[32 + t * 32 + co] = (float)a;
            bsum += a;
        }
        tab[co] = (float)bsum;
    }
    ME_HIP(hipMemcpy(ctx->arena + ctx->head_fused_off, packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
    ME_HIP(hipMemcpy(ctx->arena + ctx->head_fused_off + packed.size() * 2, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
}

// decoder.rs:101 + mod.rs:323-326: the last fusion block's out_conv (1x1, dec -> dec, bias) feeds head[0] (Conv2d 3x3, dec -> dec / 2,
// padding 1, bias) with nothing between them -- ONE 3x3 convolution of out_conv's INPUT v (the residual unit's output):
//   head0(out_conv(v))[y][x][n] = b'[n] + sum_{tap, k} v[y + ty][x + tx][k] * W''[n][tap][k],   W''[n][tap][k] = sum_c W0[n][c][tap] Wo[c][k],
//   b'[n] = b0[n] + sum_tap Cb[tap][n],   Cb[tap][n] = sum_c W0[n][c][tap] bo[c].
// The [B, dec, 768, 768] feature map and its 1x1 launch (0.9 GB of traffic) go; the residual unit's last convolution writes v as
// the zero-bordered 16-bit operand directly.  head[0] pads out_conv's OUTPUT with zeros, so a tap that falls outside the map
// must not contribute bo either: the epilogue takes that tap's Cb share out of the bias again at the image border (the v part
// vanishes by itself, v being zero-bordered) -- GemmParams::tap_bias.  Composed in f64 from the checkpoint's values, rounded
// once to the operand type.  Layout = head.0.weight's own ([dec / 2][9][dec]); tables f32 [dec / 2] + [9][dec / 2].
static void compose_features(me_ctx* ctx) {
    if (!ctx->feat_fused_off) return;
    const int64_t D = ctx->cfg.dec_dim, Co = D / 2;
    const bool to_bf16 = ctx->dtype == ME_DTYPE_BF16;
    auto to16 = [&](float f) { return to_bf16 ? float_to_bf16(f) : float_to_half(f); };
    const char* names[4] = {"decoder.fusions.0.out_conv.weight", "decoder.fusions.0.out_conv.bias", "head.0.weight", "head.0.bias"};
    bool any = false;
    for (const char* nm : names) any = any || ctx->factor_keep.count(nm);
    if (!any) return;  // nothing (re)loaded on this context: the arena's own composition (me_weights_adopt / broadcast)
    for (const char* nm : names)
        if (!ctx->slots[ctx->slot_by_name.at(nm)].loaded) return;  // finalize reports the missing one
    const std::vector<float>& Wo = fetch_factor(ctx, names[0]);  // [c][k]
    const std::vector<float>& bo = fetch_factor(ctx, names[1]);  // [c]
    const std::vector<float>& W0 = fetch_factor(ctx, names[2]);  // [n][c][tap]
    const std::vector<float>& b0 = fetch_factor(ctx, names[3]);  // [n]
    std::vector<uint16_t> packed((size_t)Co * 9 * D);
    std::vector<float> tab((size_t)10 * Co, 0.f);
    std::vector<double> acc((size_t)D);
    for (int64_t n = 0; n < Co; ++n) {
        double bsum = (double)b0[n];
        for (int t = 0; t < 9; ++t) {
            std::fill(acc.begin(), acc.end(), 0.0);
            double cb = 0.0;
            for (int64_t c = 0; c < D; ++c) {
                const double w = (double)W0[(n * D + c) * 9 + t];
                const float* row = &Wo[(size_t)c * D];
                for (int64_t k = 0; k < D; ++k) acc[k] += w * (double)row[k];
                cb += w * (double)bo[c];
            }
            uint16_t* dst = &packed[((size_t)n * 9 + t) * D];
            for (int64_t k = 0; k < D; ++k) dst[k] = to16((float)acc[k]);
            tab[(size_t)Co + t * Co + n] = (float)cb;
            bsum += cb;
        }
        tab[n] = (float)bsum;
    }
    ME_HIP(hipMemcpy(ctx->arena + ctx->feat_fused_off, packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
    ME_HIP(hipMemcpy(ctx->arena + ctx->feat_fused_off + packed.size() * 2, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
}

// ME_DTYPE_FP8: quantise qkv / proj / fc1 / fc2 of the three ViTs from the packed f16 arena (the checkpoint's values)
// to MX fp8 on the device.  Derived data: a rank that received the arena by broadcast rebuilds it itself.
void build_fp8_weights(me_ctx* ctx) {
    if (!ctx->fp8) return;
    const int64_t C = ctx->C();
    struct Item {
        const void* src;
        int64_t N, K;
        const uint8_t **w8, **ws;
    };
    std::vector<Item> items;
    for (int v = 0; v < 3; ++v)
        for (VitBlockW& b : ctx->w.vit[v].blocks) {
            items.push_back({b.qkv_w, 3 * C, C, &b.qkv_w8, &b.qkv_ws});
            items.push_back({b.fc1_w, 4 * C, C, &b.fc1_w8, &b.fc1_ws});
            items.push_back({b.fc2_w, C, 4 * C, &b.fc2_w8, &b.fc2_ws});
            items.push_back({b.proj_w, C, C, &b.proj_w8, &b.proj_ws});
        }
    size_t total = 0;
    for (const Item& it : items) total += align_up((size_t)it.N * it.K, 256) + align_up((size_t)it.N * it.K / 32, 256);
    if (!ctx->arena8 || ctx->arena8_bytes != total) {
        if (ctx->arena8) ME_HIP(hipFree(ctx->arena8));
        ctx->arena8 = nullptr;
        ME_HIP(hipMalloc((void**)&ctx->arena8, total));
        ctx->arena8_bytes = total;
    }
    size_t off = 0;
    for (const Item& it : items) {
        uint8_t* w8 = (uint8_t*)ctx->arena8 + off;
        off += align_up((size_t)it.N * it.K, 256);
        uint8_t* ws = (uint8_t*)ctx->arena8 + off;
        off += align_up((size_t)it.N * it.K / 32, 256);
        quantize_f16_to_fp8_launch(it.src, w8, ws, it.N, (int32_t)it.K, 1, ctx->stream);
        *it.w8 = w8, *it.ws = ws;
    }
    ME_HIP(hipStreamSynchronize(ctx->stream));
}

void finalize_weights(me_ctx* ctx) {
    std::string missing;
    int n = 0;
    for (const WeightSlot& s : ctx->slots)
        if (!s.loaded) {
            if (n < 8) missing += (n ? ", " : "") + s.name;
            ++n;
        }
    // LoaderError::RecorderMissing (mod.rs:241-243)
    ME_CHECK(n == 0, ME_ERR_MISSING_WEIGHT, "%d tensors missing from the checkpoint: %s%s", n,
             missing.c_str(), n > 8 ? ", ..." : "");
    compose_fusion_out(ctx);
    compose_head(ctx);
    compose_features(ctx);
    build_fp8_weights(ctx);
    ctx->finalized = true;
    ctx->drop_graph(), ++ctx->weights_generation;
}

}  // namespace me
