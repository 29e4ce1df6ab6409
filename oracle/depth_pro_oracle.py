"""CPU oracle for the Depth Pro forward pass — TEST INFRASTRUCTURE, NOT THE PRODUCT.

A restatement of the reference's algorithm (zlogic/matrix-eyes v0.1.7, src/depth_pro/*.rs) in
PyTorch-CPU fp32 (fp64 on request), written from the Rust text; every function cites the lines it
follows.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

PARITY UNPINNED: the reference has no tests, fixtures or golden vectors, its
arithmetic lives in the un-vendored third-party crate `burn` 0.21.0 (Cargo.toml:11-12,
Cargo.lock:400) and no Rust toolchain exists in the build image, so this restatement could not be
checked against outputs of the reference itself.  It is pinned instead by op-level known-answer
tests against torch.nn.functional and by the geometric invariants of split/merge
(tests/test_oracle_depth_pro.py).  Third-party semantics that had to be assumed:
LayerNorm eps 1e-5 with biased variance; bilinear interpolate = align_corners=True; exact-erf GELU;
max-subtracted softmax.  Both are parameters here.

Weights are a dict name -> tensor under the PyTorch checkpoint names/layouts;
the reference's PyTorchToBurnAdapter transposes Linear weights to [in,out] and computes x.W
(mod.rs:234), which equals F.linear with the [out,in] layout.
"""
import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F


@dataclass
class OracleConfig:
    grid: int = 24                       # vit.rs:17-18
    embed_dim: int = 1024                # vit.rs:19
    num_heads: int = 16                  # vit.rs:355
    depth: int = 24                      # vit.rs:353
    tap_blocks: Tuple[int, int] = (5, 11)                       # encoder.rs:227
    enc_dims: Tuple[int, int, int, int] = (256, 512, 1024, 1024)  # mod.rs:262
    dec_dim: int = 256                   # mod.rs:263
    head_dims: Tuple[int, int] = (32, 1)   # mod.rs:310
    ln_eps: float = 1e-5
    align_corners: bool = True
    dtype: torch.dtype = torch.float32

    @property
    def window(self):
        return 16 * self.grid


def _w(weights: Dict[str, torch.Tensor], name: str, cfg: OracleConfig) -> torch.Tensor:
    return weights[name].to(cfg.dtype)


# ---------------------------------------------------------------------------------------------
# vit.rs
# ---------------------------------------------------------------------------------------------
def layer_norm(x, weight, bias, eps):
    """burn nn::LayerNorm::forward: var_mean_bias over the last dim, (x-mean)/sqrt(var+eps)*g+b."""
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * weight + bias


def gelu(x):
    """burn activation::gelu: x * 0.5 * (1 + erf(x / sqrt(2)))  (vit.rs:121)"""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def attention_forward(xs, weights, p, cfg):
    """vit.rs:58-75 Attention::forward"""
    b, n, c = xs.shape
    h = cfg.num_heads
    qkv = F.linear(xs, _w(weights, p + "qkv.weight", cfg), _w(weights, p + "qkv.bias", cfg))
    qkv = qkv.reshape(b, n, 3, h, c // h).permute(2, 0, 3, 1, 4)       # :63-64
    q, k, v = qkv[0], qkv[1], qkv[2]
    scale = 1.0 / math.sqrt(c // h)                                      # :47
    q = q * scale                                                        # :69
    attn = torch.softmax(q @ k.transpose(3, 2), dim=3)                   # :72
    out = (attn @ v).transpose(1, 2).reshape(b, n, c)                    # :73
    return F.linear(out, _w(weights, p + "proj.weight", cfg), _w(weights, p + "proj.bias", cfg))


def block_forward(xs, weights, p, cfg):
    """vit.rs:163-170 Block::forward (pre-LN, LayerScale :93-95, Mlp :119-123)"""
    residual = xs
    y = layer_norm(xs, _w(weights, p + "norm1.weight", cfg), _w(weights, p + "norm1.bias", cfg), cfg.ln_eps)
    y = attention_forward(y, weights, p + "attn.", cfg) * _w(weights, p + "ls1.gamma", cfg)
    xs = y + residual
    residual = xs
    y = layer_norm(xs, _w(weights, p + "norm2.weight", cfg), _w(weights, p + "norm2.bias", cfg), cfg.ln_eps)
    y = F.linear(y, _w(weights, p + "mlp.fc1.weight", cfg), _w(weights, p + "mlp.fc1.bias", cfg))
    y = gelu(y)
    y = F.linear(y, _w(weights, p + "mlp.fc2.weight", cfg), _w(weights, p + "mlp.fc2.bias", cfg))
    y = y * _w(weights, p + "ls2.gamma", cfg)
    return y + residual


def patch_embed_forward(xs, weights, p, cfg):
    """vit.rs:210-223 PatchEmbed::forward"""
    _, _, h, w = xs.shape
    if h % 16 or w % 16:   # :213-218
        raise ValueError(f"image {h}x{w} is not a multiple of the patch size 16")
    y = F.conv2d(xs, _w(weights, p + "proj.weight", cfg), _w(weights, p + "proj.bias", cfg), stride=16)
    b, c, hh, ww = y.shape
    return y.reshape(b, c, hh * ww).transpose(1, 2)


def prepare_tokens_with_mask(xs, weights, p, cfg):
    """vit.rs:287-295"""
    b = xs.shape[0]
    y = patch_embed_forward(xs, weights, p + "patch_embed.", cfg)
    cls = _w(weights, p + "cls_token", cfg).expand(b, -1, -1)
    y = torch.cat([cls, y], dim=1)
    pos = _w(weights, p + "pos_embed", cfg)
    if pos.shape[1] != y.shape[1]:   # :281-283
        raise ValueError("pos_embed interpolation is not implemented")
    return y + pos


def vit_forward_features(xs, weights, p, cfg, intermediate_blocks: Sequence[int] = ()):
    """vit.rs:328-346 forward_features (+ :297-326 get_intermediate_layers_not_chunked)"""
    y = prepare_tokens_with_mask(xs, weights, p, cfg)
    outputs = []
    for i in range(cfg.depth):
        y = block_forward(y, weights, f"{p}blocks.{i}.", cfg)
        if i in intermediate_blocks:
            outputs.append(y.clone())
    if len(outputs) != len(intermediate_blocks):   # :318-324
        raise ValueError(f"only {len(outputs)} / {len(intermediate_blocks)} blocks found")
    final = layer_norm(y, _w(weights, p + "norm.weight", cfg), _w(weights, p + "norm.bias", cfg), cfg.ln_eps)
    return final, outputs


# ---------------------------------------------------------------------------------------------
# encoder.rs
# ---------------------------------------------------------------------------------------------
def interpolate_bilinear(x, out_h, out_w, align_corners=True):
    """burn module::interpolate(..., Bilinear) (encoder.rs:128-137, fov.rs:53).

    align_corners=True (assumed Burn semantics): src = dst*(in-1)/(out-1), weights
    from the fractional part, evaluated in f64 like burn-ndarray and rounded to the tensor dtype.
    align_corners=False: half-pixel centres (torch semantics)."""
    if not align_corners:
        return F.interpolate(x, size=(out_h, out_w), mode="bilinear", align_corners=False)
    _, _, in_h, in_w = x.shape
    xd = x.to(torch.float64)
    ry = (in_h - 1) / max(out_h - 1, 1)
    rx = (in_w - 1) / max(out_w - 1, 1)
    fy = torch.arange(out_h, dtype=torch.float64) * ry
    fx = torch.arange(out_w, dtype=torch.float64) * rx
    y0 = fy.floor().clamp(max=in_h - 1).long()
    x0 = fx.floor().clamp(max=in_w - 1).long()
    y1 = (y0 + 1).clamp(max=in_h - 1)
    x1 = (x0 + 1).clamp(max=in_w - 1)
    wy = (fy - y0.to(torch.float64)).view(1, 1, -1, 1)
    wx = (fx - x0.to(torch.float64)).view(1, 1, 1, -1)
    a = xd[:, :, y0][:, :, :, x0]
    b = xd[:, :, y0][:, :, :, x1]
    c = xd[:, :, y1][:, :, :, x0]
    d = xd[:, :, y1][:, :, :, x1]
    out = a * (1 - wx) * (1 - wy) + b * wx * (1 - wy) + c * (1 - wx) * wy + d * wx * wy
    return out.to(x.dtype)


def create_pyramid(x, cfg):
    """encoder.rs:125-140"""
    _, _, h, w = x.shape
    x1 = interpolate_bilinear(x, w // 2, h // 2, cfg.align_corners)   # [w/2, h/2] as written
    x2 = interpolate_bilinear(x, w // 4, h // 4, cfg.align_corners)
    return x, x1, x2


def split(x, overlap_div, patch_size):
    """encoder.rs:142-156 (PATCH_SIZE 384 generalised to the window size)"""
    stride = patch_size - patch_size // overlap_div
    image_size = x.shape[3]
    patches = []
    for j in range(0, image_size - patch_size + 1, stride):
        chunk = x[:, :, j:j + patch_size, :]
        for i in range(0, image_size - patch_size + 1, stride):
            patches.append(chunk[:, :, :, i:i + patch_size])
    return torch.cat(patches, dim=0)


def merge(x, batch_size, padding):
    """encoder.rs:158-189"""
    b, c, h, w = x.shape
    steps = int(math.sqrt(b // batch_size))
    rows = []
    for j in range(steps):
        row = []
        for i in range(steps):
            idx = j * steps + i
            h0, h1, w0, w1 = 0, h, 0, w
            if j > 0:
                h0 = padding
            if i > 0:
                w0 = padding
            if j < steps - 1:
                h1 = h - padding
            if i < steps - 1:
                w1 = w - padding
            row.append(x[batch_size * idx:batch_size * (idx + 1), :, h0:h1, w0:w1])
        rows.append(torch.cat(row, dim=3))
    return torch.cat(rows, dim=2)


def reshape_feature(emb, width, height, cls_token_offset):
    """encoder.rs:191-208"""
    b, hw, c = emb.shape
    if cls_token_offset > 0:
        emb = emb[:, cls_token_offset:, :]
    return emb.reshape(b, height, width, c).permute(0, 3, 1, 2)


def _upsample_block(x, weights, p, n_convt, cfg):
    """encoder.rs:210-216 forward_seq over init_project_upsample_block (:85-118): 1x1 conv (no
    bias) then n ConvTranspose2d(2,2,stride 2, no bias)."""
    y = F.conv2d(x, _w(weights, p + "0.weight", cfg))
    for i in range(n_convt):
        y = F.conv_transpose2d(y, _w(weights, f"{p}{i + 1}.weight", cfg), stride=2)
    return y


def encoder_forward_encodings(x, weights, cfg) -> List[torch.Tensor]:
    """encoder.rs:218-335 DepthProEncoder::forward_encodings"""
    g = cfg.grid
    batch = x.shape[0]
    x0, x1, x2 = create_pyramid(x, cfg)                                   # :234
    x0_patches = split(x0, 4, cfg.window)                                 # :238
    x1_patches = split(x1, 2, cfg.window)                                 # :240
    x2_patches = x2
    n0, n1, n2 = x0_patches.shape[0], x1_patches.shape[0], x2_patches.shape[0]
    pyramid = torch.cat([x0_patches, x1_patches, x2_patches], dim=0)      # :249-250
    enc, taps = vit_forward_features(pyramid, weights, "encoder.patch_encoder.", cfg,
                                     list(cfg.tap_blocks))                # :254-256
    pad0, pad1 = g // 8, g // 4            # 3 and 6 at grid 24 (:270,279,291,293)
    enc = reshape_feature(enc, g, g, 1)                                   # :263
    lat0 = merge(reshape_feature(taps[0], g, g, 1)[:batch * 25], batch, pad0)   # :266-271
    lat1 = merge(reshape_feature(taps[1], g, g, 1)[:batch * 25], batch, pad0)   # :274-280
    x0_enc, x1_enc, x2_enc = torch.split(enc, [n0, n1, n2], dim=0)        # :285-288
    x0_feat = merge(x0_enc, batch, pad0)                                  # :291
    x1_feat = merge(x1_enc, batch, pad1)                                  # :293
    x2_feat = x2_enc
    glob, _ = vit_forward_features(x2_patches, weights, "encoder.image_encoder.", cfg, [])  # :298-300
    glob = reshape_feature(glob, g, g, 1)                                 # :303
    lat0 = _upsample_block(lat0, weights, "encoder.upsample_latent0.", 3, cfg)   # :307
    lat1 = _upsample_block(lat1, weights, "encoder.upsample_latent1.", 2, cfg)   # :309
    x0_feat = _upsample_block(x0_feat, weights, "encoder.upsample0.", 1, cfg)    # :312
    x1_feat = _upsample_block(x1_feat, weights, "encoder.upsample1.", 1, cfg)    # :314
    x2_feat = _upsample_block(x2_feat, weights, "encoder.upsample2.", 1, cfg)    # :316
    glob = F.conv_transpose2d(glob, _w(weights, "encoder.upsample_lowres.weight", cfg),
                              _w(weights, "encoder.upsample_lowres.bias", cfg), stride=2)   # :320
    glob = F.conv2d(torch.cat([x2_feat, glob], dim=1), _w(weights, "encoder.fuse_lowres.weight", cfg),
                    _w(weights, "encoder.fuse_lowres.bias", cfg))         # :323-325
    return [lat0, lat1, x0_feat, x1_feat, glob]                           # :328-334


# ---------------------------------------------------------------------------------------------
# decoder.rs
# ---------------------------------------------------------------------------------------------
def _rcu(x, weights, p, cfg):
    """decoder.rs:35-44 ResidualConvUnit::forward (PyTorch Sequential indices 1 and 3: the convs)"""
    out = x
    for idx in ("1", "3"):
        out = F.relu(out)
        out = F.conv2d(out, _w(weights, f"{p}residual.{idx}.weight", cfg),
                       _w(weights, f"{p}residual.{idx}.bias", cfg), padding=1)
    return x + out


def _fusion(x0, x1, weights, p, has_deconv, cfg):
    """decoder.rs:84-102 FeatureFusionBlock::forward"""
    out = x0
    if x1 is not None:
        out = x0 + _rcu(x1, weights, p + "resnet1.", cfg)
    out = _rcu(out, weights, p + "resnet2.", cfg)
    if has_deconv:
        out = F.conv_transpose2d(out, _w(weights, p + "deconv.weight", cfg), stride=2)
    return F.conv2d(out, _w(weights, p + "out_conv.weight", cfg), _w(weights, p + "out_conv.bias", cfg))


def decoder_forward(encodings: List[torch.Tensor], weights, cfg):
    """decoder.rs:153-208 MultiresConvDecoder::forward -> (features, lowres_features)"""
    if len(encodings) != 5:   # :161-165
        raise ValueError(f"got encoder output levels {len(encodings)}, expected levels 5")
    feats = F.conv2d(encodings[4], _w(weights, "decoder.convs.4.weight", cfg), padding=1)   # :171-176
    lowres = feats.clone()                                                                 # :178
    feats = _fusion(feats, None, weights, "decoder.fusions.4.", True, cfg)                 # :179-183
    for i in (3, 2, 1, 0):                                                                 # :188-205
        enc = encodings[i]
        if i >= 1:
            enc = F.conv2d(enc, _w(weights, f"decoder.convs.{i}.weight", cfg), padding=1)
        feats = _fusion(feats, enc, weights, f"decoder.fusions.{i}.", i != 0, cfg)
    return feats, lowres


# ---------------------------------------------------------------------------------------------
# mod.rs head, fov.rs, mod.rs extract_depth
# ---------------------------------------------------------------------------------------------
def head_forward(features, weights, cfg):
    """mod.rs:323-338 -> canonical inverse depth [B, S, S] (the reference squeezes batch 1)"""
    y = F.conv2d(features, _w(weights, "head.0.weight", cfg), _w(weights, "head.0.bias", cfg), padding=1)
    y = F.conv_transpose2d(y, _w(weights, "head.1.weight", cfg), _w(weights, "head.1.bias", cfg), stride=2)
    y = F.conv2d(y, _w(weights, "head.2.weight", cfg), _w(weights, "head.2.bias", cfg), padding=1)
    y = F.relu(y)
    y = F.conv2d(y, _w(weights, "head.4.weight", cfg), _w(weights, "head.4.bias", cfg))
    y = F.relu(y)
    return y[:, 0]


def fov_forward(x, lowres_feature, weights, cfg):
    """fov.rs:40-88 FOVNetwork::forward -> fov_deg [B]"""
    _, _, h, w = x.shape
    x = interpolate_bilinear(x, w // 4, h // 4, cfg.align_corners)                          # :53
    y, _ = vit_forward_features(x, weights, "fov.encoder.0.", cfg, [])                      # :57-61
    y = F.linear(y, _w(weights, "fov.encoder.1.weight", cfg), _w(weights, "fov.encoder.1.bias", cfg))  # :63
    y = y[:, 1:, :].permute(0, 2, 1)                                                        # :66-67
    low = F.conv2d(lowres_feature, _w(weights, "fov.downsample.0.weight", cfg),
                   _w(weights, "fov.downsample.0.bias", cfg), stride=2, padding=1)          # :70
    low = F.relu(low)                                                                       # :72
    y = y.reshape(low.shape) + low                                                          # :74
    y = F.relu(F.conv2d(y, _w(weights, "fov.head.0.weight", cfg), _w(weights, "fov.head.0.bias", cfg),
                        stride=2, padding=1))                                               # :77-79
    y = F.relu(F.conv2d(y, _w(weights, "fov.head.2.weight", cfg), _w(weights, "fov.head.2.bias", cfg),
                        stride=2, padding=1))                                               # :81-83
    y = F.conv2d(y, _w(weights, "fov.head.4.weight", cfg), _w(weights, "fov.head.4.bias", cfg))   # :85
    return y.reshape(-1)                                                                    # :87


def f_norm_from_fov(fov_deg: float) -> float:
    """mod.rs:358: (0.5 * (fov_deg * PI / 180.0)).tan() / 0.5, in f32 (quirk Q1, as written)"""
    import numpy as np
    d = np.float32(fov_deg)
    return float(np.tan(np.float32(0.5) * (d * np.float32(np.pi) / np.float32(180.0))) / np.float32(0.5))


def extract_depth(img, f_norm: Optional[float], weights, cfg, return_parts=False):
    """mod.rs:251-363 DepthProModelLoader::extract_depth for one image [1,3,S,S] (or a batch, each
    image independently) -> inverse depth [B,S,S] (+ fov_deg [B] or None)"""
    with torch.no_grad():
        img = img.to(cfg.dtype)
        encodings = encoder_forward_encodings(img, weights, cfg)       # :276-288
        features, lowres = decoder_forward(encodings, weights, cfg)    # :291-304
        canonical = head_forward(features, weights, cfg)               # :307-338
        fov_deg = None
        if f_norm is None:                                             # :340-359
            fov_deg = fov_forward(img, lowres, weights, cfg)
            fn = torch.tensor([f_norm_from_fov(float(v)) for v in fov_deg], dtype=cfg.dtype)
        else:
            fn = torch.full((img.shape[0],), float(f_norm), dtype=cfg.dtype)
        inv = (canonical / fn.view(-1, 1, 1)).clamp(1e-4, 1e4)         # :361-362
        if return_parts:
            return inv, fov_deg, dict(encodings=encodings, features=features, lowres=lowres,
                                      canonical=canonical)
        return inv, fov_deg


def preprocess_u8(rgb):
    """reconstruction.rs:116-124: u8 HWC [B,S,S,3] -> f32 [B,3,S,S], (x/255 - 0.5)/0.5 in f32"""
    x = torch.as_tensor(rgb).to(torch.float32).permute(0, 3, 1, 2) / 255.0
    return ((x - 0.5) / 0.5).contiguous()
