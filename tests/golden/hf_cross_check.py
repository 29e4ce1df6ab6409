"""Cross-check of the CPU oracle's WIRING against an independent implementation (build container only, CPU).

The reference holds no fixtures and cannot be built here, so the oracle is "parity unpinned" (DESIGN.md section 3).
The one independent implementation of Depth Pro in this image is Hugging Face transformers'
`DepthProForDepthEstimation` (transformers/models/depth_pro/modeling_depth_pro.py), written from Apple's
PyTorch code, not from the Rust reference.  This script loads ONE synthetic checkpoint into both -- converting
the reference's PyTorch key names to HF's -- and compares stage outputs.  It does not pin the
oracle's arithmetic to the reference (Burn's LayerNorm eps and bilinear convention stay assumptions, SURVEY
App. D; HF is set to the oracle's choices: eps 1e-5, align_corners=True), but it is independent evidence for
everything else: checkpoint key mapping, qkv column order, head split, window order of split / merge and its
paddings, which block each tap is, the upsample / fuse / decoder / head / FOV wiring.

Writes tests/golden/hf_cross_check.json (the figures tests/test_oracle_depth_pro.py asserts on).
    python tests/golden/hf_cross_check.py
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import matrix_eyes_amd as m                                      # noqa: E402
from matrix_eyes_amd.synthetic import synthetic_checkpoint, synthetic_images  # noqa: E402
from oracle import depth_pro_oracle as O                          # noqa: E402


def hf_config(cfg):
    from transformers import DepthProConfig
    vit = dict(hidden_size=cfg.embed_dim, num_hidden_layers=cfg.depth, num_attention_heads=cfg.num_heads,
               image_size=cfg.window, patch_size=16, layer_norm_eps=cfg.ln_eps, mlp_ratio=4, use_swiglu_ffn=False,
               qkv_bias=True, layerscale_value=1.0, model_type="dinov2")
    e0, e1, e2, e3 = cfg.enc_dims
    assert e0 == cfg.dec_dim, "HF hard-codes intermediate_dims = fusion_hidden_size for the deeper hook"
    return DepthProConfig(
        fusion_hidden_size=cfg.dec_dim, patch_size=cfg.window,
        intermediate_hook_ids=[cfg.tap_blocks[1], cfg.tap_blocks[0]], intermediate_feature_dims=[e0, cfg.dec_dim],
        scaled_images_ratios=[0.25, 0.5, 1], scaled_images_overlap_ratios=[0.0, 0.5, 0.25],
        scaled_images_feature_dims=[e3, e2, e1], merge_padding_value=cfg.grid // 8, use_fov_model=True,
        num_fov_head_layers=2, patch_model_config=dict(vit), image_model_config=dict(vit), fov_model_config=dict(vit))


def convert_vit(w, src, dst, cfg, out):
    """reference ViT keys (vit.rs field names after mod.rs:185-210) -> HF Dinov2Model keys"""
    C = cfg.embed_dim
    out[dst + "embeddings.cls_token"] = w[src + "cls_token"]
    out[dst + "embeddings.mask_token"] = torch.zeros(1, C)
    out[dst + "embeddings.position_embeddings"] = w[src + "pos_embed"]
    out[dst + "embeddings.patch_embeddings.projection.weight"] = w[src + "patch_embed.proj.weight"]
    out[dst + "embeddings.patch_embeddings.projection.bias"] = w[src + "patch_embed.proj.bias"]
    for i in range(cfg.depth):
        s, d = f"{src}blocks.{i}.", f"{dst}encoder.layer.{i}."
        qkv_w, qkv_b = w[s + "attn.qkv.weight"], w[s + "attn.qkv.bias"]
        for j, n in enumerate(("query", "key", "value")):          # vit.rs:63-68: columns [3][heads][64]
            out[f"{d}attention.attention.{n}.weight"] = qkv_w[j * C:(j + 1) * C]
            out[f"{d}attention.attention.{n}.bias"] = qkv_b[j * C:(j + 1) * C]
        out[d + "attention.output.dense.weight"] = w[s + "attn.proj.weight"]
        out[d + "attention.output.dense.bias"] = w[s + "attn.proj.bias"]
        out[d + "layer_scale1.lambda1"] = w[s + "ls1.gamma"]
        out[d + "layer_scale2.lambda1"] = w[s + "ls2.gamma"]
        for n in ("norm1", "norm2"):
            out[f"{d}{n}.weight"], out[f"{d}{n}.bias"] = w[f"{s}{n}.weight"], w[f"{s}{n}.bias"]
        for n in ("fc1", "fc2"):
            out[f"{d}mlp.{n}.weight"], out[f"{d}mlp.{n}.bias"] = w[f"{s}mlp.{n}.weight"], w[f"{s}mlp.{n}.bias"]
    out[dst + "layernorm.weight"], out[dst + "layernorm.bias"] = w[src + "norm.weight"], w[src + "norm.bias"]


def convert(w, cfg):
    out = {}
    convert_vit(w, "encoder.patch_encoder.", "depth_pro.encoder.patch_encoder.model.", cfg, out)
    convert_vit(w, "encoder.image_encoder.", "depth_pro.encoder.image_encoder.model.", cfg, out)
    convert_vit(w, "fov.encoder.0.", "fov_model.fov_encoder.model.", cfg, out)
    up = "depth_pro.neck.feature_upsample."
    # encoder.rs:48-71 <-> DepthProFeatureUpsample: scaled_images follow ratios [1/4, 1/2, 1]
    for ours, theirs, layers in (("upsample2", "scaled_images.0", 1), ("upsample1", "scaled_images.1", 1),
                                 ("upsample0", "scaled_images.2", 1), ("upsample_latent1", "intermediate.0", 2),
                                 ("upsample_latent0", "intermediate.1", 3)):
        for k in range(layers + 1):
            out[f"{up}{theirs}.layers.{k}.weight"] = w[f"encoder.{ours}.{k}.weight"]
    out[up + "image_block.layers.0.weight"] = w["encoder.upsample_lowres.weight"]
    out[up + "image_block.layers.0.bias"] = w["encoder.upsample_lowres.bias"]
    out["depth_pro.neck.fuse_image_with_low_res.weight"] = w["encoder.fuse_lowres.weight"]
    out["depth_pro.neck.fuse_image_with_low_res.bias"] = w["encoder.fuse_lowres.bias"]
    # decoder.rs:123-131 convs[i-1] for level i <-> projections (lowest resolution first)
    for j in range(4):
        out[f"depth_pro.neck.feature_projection.projections.{j}.weight"] = w[f"decoder.convs.{4 - j}.weight"]
    # decoder.rs:133-146 fusions[i] (level i) <-> fusion_stage.intermediate[4 - i] / final (level 0)
    for i in range(5):
        dst = "fusion_stage.final." if i == 0 else f"fusion_stage.intermediate.{4 - i}."
        src = f"decoder.fusions.{i}."
        for ours, theirs in (("resnet1", "residual_layer1"), ("resnet2", "residual_layer2")):
            for a, b in (("1", "convolution1"), ("3", "convolution2")):
                out[f"{dst}{theirs}.{b}.weight"] = w[f"{src}{ours}.residual.{a}.weight"]
                out[f"{dst}{theirs}.{b}.bias"] = w[f"{src}{ours}.residual.{a}.bias"]
        if i != 0:
            out[dst + "deconv.weight"] = w[src + "deconv.weight"]
        out[dst + "projection.weight"], out[dst + "projection.bias"] = w[src + "out_conv.weight"], w[src + "out_conv.bias"]
    for k in (0, 1, 2, 4):
        out[f"head.layers.{k}.weight"], out[f"head.layers.{k}.bias"] = w[f"head.{k}.weight"], w[f"head.{k}.bias"]
    out["fov_model.fov_encoder.neck.weight"], out["fov_model.fov_encoder.neck.bias"] = w["fov.encoder.1.weight"], w["fov.encoder.1.bias"]
    out["fov_model.conv.weight"], out["fov_model.conv.bias"] = w["fov.downsample.0.weight"], w["fov.downsample.0.bias"]
    for k in (0, 2, 4):
        out[f"fov_model.head.layers.{k}.weight"] = w[f"fov.head.{k}.weight"]
        out[f"fov_model.head.layers.{k}.bias"] = w[f"fov.head.{k}.bias"]
    return {k: torch.as_tensor(v).float() for k, v in out.items()}


def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm())


def main():
    from transformers import DepthProForDepthEstimation
    from transformers.models.depth_pro import modeling_depth_pro as hf
    # the real model's dims have enc_dims[0] == dec_dim; HF relies on it, so the test geometry keeps it
    cfg = m.ModelConfig(grid=8, embed_dim=128, num_heads=2, depth=4, tap_blocks=(1, 2),
                        enc_dims=(256, 128, 192, 320), dec_dim=256, head_dims=(32, 1))
    w = {k: torch.as_tensor(v) for k, v in synthetic_checkpoint(cfg).items()}
    model = DepthProForDepthEstimation(hf_config(cfg), use_fov_model=True).eval()
    state = convert(w, cfg)
    missing, unexpected = model.load_state_dict(state, strict=False)
    assert not unexpected, unexpected
    assert all("residual_layer1" in k and "intermediate.0." in k for k in missing), missing  # decoder.rs:171-183: unused
    # HF interpolates with align_corners=False (Apple's convention); the oracle assumes Burn's historical
    # align_corners=True.  Put HF on the oracle's convention: same-size calls are the identity
    # under both, so only the pyramid / FOV down-scalings change.
    real = hf.F.interpolate

    def interpolate(x, size=None, scale_factor=None, mode="nearest", align_corners=None, **kw):
        return real(x, size=size, scale_factor=scale_factor, mode=mode, align_corners=True if mode == "bilinear" else align_corners, **kw)
    hf.F.interpolate = interpolate
    img = O.preprocess_u8(synthetic_images(1, cfg.img_size))
    ocfg = O.OracleConfig(grid=cfg.grid, embed_dim=cfg.embed_dim, num_heads=cfg.num_heads, depth=cfg.depth,
                          tap_blocks=cfg.tap_blocks, enc_dims=cfg.enc_dims, dec_dim=cfg.dec_dim, head_dims=cfg.head_dims,
                          ln_eps=cfg.ln_eps, align_corners=True)
    with torch.no_grad():
        out = model(pixel_values=img, return_dict=True)
        neck = model.depth_pro(pixel_values=img, return_dict=True).features
        fused = model.fusion_stage(list(neck))
    inv, fov, parts = O.extract_depth(img, None, w, ocfg, return_parts=True)
    res = {
        "config": dict(grid=cfg.grid, embed_dim=cfg.embed_dim, depth=cfg.depth, enc_dims=list(cfg.enc_dims), dec_dim=cfg.dec_dim),
        "transformers": __import__("transformers").__version__,
        "canonical_inverse_depth_rel_l2": rel(out.predicted_depth, parts["canonical"]),
        "features_rel_l2": rel(fused[-1], parts["features"]),
        "lowres_features_rel_l2": rel(neck[0], parts["lowres"]),
        "fov_deg": [float(out.field_of_view[0]), float(fov[0])],
    }
    print(json.dumps(res, indent=1))
    json.dump(res, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "hf_cross_check.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
