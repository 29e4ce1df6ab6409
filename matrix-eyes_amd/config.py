"""Model geometry (mirror of `me_model_config`) and the table of checkpoint tensors the model
expects (mirror of csrc/weights.hip; reference src/depth_pro/*.rs module definitions and the
key-remap rules of mod.rs:185-210)."""
from dataclasses import dataclass, field
from typing import List, Tuple

from ._lib import CModelConfig


@dataclass
class ModelConfig:
    grid: int = 24                      # vit.rs:17-18: IMG_SIZE 384 / PATCH_SIZE 16
    embed_dim: int = 1024               # vit.rs:19
    num_heads: int = 16                 # vit.rs:355
    depth: int = 24                     # vit.rs:353
    tap_blocks: Tuple[int, int] = (5, 11)                       # encoder.rs:227
    enc_dims: Tuple[int, int, int, int] = (256, 512, 1024, 1024)  # mod.rs:262
    dec_dim: int = 256                  # mod.rs:263
    head_dims: Tuple[int, int] = (32, 1)  # mod.rs:310
    ln_eps: float = 1e-5                # Burn LayerNormConfig default
    align_corners: bool = True          # Burn bilinear
    split_operands: int = 3             # me_model_config.split_operands: hi + lo operand stages (bit mask)
    fp8_linears: int = 0                # me_model_config.fp8_linears: fp8 contexts, which linears run on fp8 (0 = default)

    @property
    def window(self) -> int:
        return 16 * self.grid

    @property
    def img_size(self) -> int:          # mod.rs:33 IMG_SIZE = vit::IMG_SIZE * 4
        return 64 * self.grid

    @property
    def tokens(self) -> int:
        return self.grid * self.grid + 1

    def to_c(self) -> CModelConfig:
        c = CModelConfig()
        c.grid, c.embed_dim, c.num_heads, c.depth = self.grid, self.embed_dim, self.num_heads, self.depth
        c.tap_blocks[0], c.tap_blocks[1] = self.tap_blocks
        for i in range(4):
            c.enc_dims[i] = self.enc_dims[i]
        c.dec_dim = self.dec_dim
        c.head_dims[0], c.head_dims[1] = self.head_dims
        c.ln_eps = self.ln_eps
        c.align_corners = 1 if self.align_corners else 0
        c.split_operands = self.split_operands
        c.fp8_linears = self.fp8_linears
        return c

    @staticmethod
    def tiny() -> "ModelConfig":
        """Same code paths at a size the CPU oracle finishes in seconds: 512x512 input,
        35 windows of 128x128, 65 tokens, ViT of 4 blocks x 128 channels."""
        return ModelConfig(grid=8, embed_dim=128, num_heads=2, depth=4, tap_blocks=(1, 2),
                           enc_dims=(64, 128, 128, 128), dec_dim=256, head_dims=(32, 1))


def _vit(prefix: str, cfg: ModelConfig) -> List[Tuple[str, Tuple[int, ...], str]]:
    C, T = cfg.embed_dim, cfg.tokens
    out = [
        (prefix + "cls_token", (1, 1, C), "embed"),
        (prefix + "pos_embed", (1, T, C), "embed"),
        (prefix + "patch_embed.proj.weight", (C, 3, 16, 16), "conv"),
        (prefix + "patch_embed.proj.bias", (C,), "bias"),
    ]
    for i in range(cfg.depth):
        b = f"{prefix}blocks.{i}."
        out += [
            (b + "norm1.weight", (C,), "ln_weight"), (b + "norm1.bias", (C,), "ln_bias"),
            (b + "attn.qkv.weight", (3 * C, C), "linear"), (b + "attn.qkv.bias", (3 * C,), "bias"),
            (b + "attn.proj.weight", (C, C), "linear"), (b + "attn.proj.bias", (C,), "bias"),
            (b + "ls1.gamma", (C,), "layer_scale"),
            (b + "norm2.weight", (C,), "ln_weight"), (b + "norm2.bias", (C,), "ln_bias"),
            (b + "mlp.fc1.weight", (4 * C, C), "linear"), (b + "mlp.fc1.bias", (4 * C,), "bias"),
            (b + "mlp.fc2.weight", (C, 4 * C), "linear"), (b + "mlp.fc2.bias", (C,), "bias"),
            (b + "ls2.gamma", (C,), "layer_scale"),
        ]
    out += [(prefix + "norm.weight", (C,), "ln_weight"), (prefix + "norm.bias", (C,), "ln_bias")]
    return out


def _upsample(prefix, cfg, dim_out, layers, dim_int):
    out = [(prefix + "0.weight", (dim_int, cfg.embed_dim, 1, 1), "conv")]
    for i in range(layers):
        cin = dim_int if i == 0 else dim_out
        out.append((f"{prefix}{i + 1}.weight", (cin, dim_out, 2, 2), "convt"))
    return out


def expected_weights(cfg: ModelConfig) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(name, PyTorch-layout shape, kind) for every tensor of the checkpoint, in the library's
    order.  kind drives the synthetic initialiser only."""
    C, dec = cfg.embed_dim, cfg.dec_dim
    e0, e1, e2, e3 = cfg.enc_dims
    out = _vit("encoder.patch_encoder.", cfg) + _vit("encoder.image_encoder.", cfg)
    out += _upsample("encoder.upsample_latent0.", cfg, dec, 3, e0)   # encoder.rs:48-54
    out += _upsample("encoder.upsample_latent1.", cfg, e0, 2, e0)    # encoder.rs:55-56
    out += _upsample("encoder.upsample0.", cfg, e1, 1, e1)
    out += _upsample("encoder.upsample1.", cfg, e2, 1, e2)
    out += _upsample("encoder.upsample2.", cfg, e3, 1, e3)
    out += [
        ("encoder.upsample_lowres.weight", (C, e3, 2, 2), "convt"),
        ("encoder.upsample_lowres.bias", (e3,), "bias"),
        ("encoder.fuse_lowres.weight", (e3, 2 * e3, 1, 1), "conv"),
        ("encoder.fuse_lowres.bias", (e3,), "bias"),
    ]
    dims_enc = (dec, e0, e1, e2, e3)                                  # mod.rs:293-295
    for i in range(1, 5):                                             # decoder.rs:132-139
        out.append((f"decoder.convs.{i}.weight", (dec, dims_enc[i], 3, 3), "conv"))
    for i in range(5):                                                # decoder.rs:141-143
        f = f"decoder.fusions.{i}."
        for rn in ("resnet1", "resnet2"):
            for idx in ("1", "3"):
                out.append((f"{f}{rn}.residual.{idx}.weight", (dec, dec, 3, 3), "conv"))
                out.append((f"{f}{rn}.residual.{idx}.bias", (dec,), "bias"))
        if i != 0:
            out.append((f + "deconv.weight", (dec, dec, 2, 2), "convt"))
        out.append((f + "out_conv.weight", (dec, dec, 1, 1), "conv"))
        out.append((f + "out_conv.bias", (dec,), "bias"))
    h0, h1 = cfg.head_dims
    out += [                                                          # mod.rs:57-97
        ("head.0.weight", (dec // 2, dec, 3, 3), "conv"), ("head.0.bias", (dec // 2,), "bias"),
        ("head.1.weight", (dec // 2, dec // 2, 2, 2), "convt"), ("head.1.bias", (dec // 2,), "bias"),
        ("head.2.weight", (h0, dec // 2, 3, 3), "conv"), ("head.2.bias", (h0,), "bias"),
        ("head.4.weight", (h1, h0, 1, 1), "conv"), ("head.4.bias", (h1,), "head_bias"),
    ]
    out += _vit("fov.encoder.0.", cfg)                                # fov.rs:95-128
    k = cfg.grid // 4
    out += [
        ("fov.encoder.1.weight", (dec // 2, C), "linear"), ("fov.encoder.1.bias", (dec // 2,), "bias"),
        ("fov.downsample.0.weight", (dec // 2, dec, 3, 3), "conv"),
        ("fov.downsample.0.bias", (dec // 2,), "bias"),
        ("fov.head.0.weight", (dec // 4, dec // 2, 3, 3), "conv"), ("fov.head.0.bias", (dec // 4,), "bias"),
        ("fov.head.2.weight", (dec // 8, dec // 4, 3, 3), "conv"), ("fov.head.2.bias", (dec // 8,), "bias"),
        ("fov.head.4.weight", (1, dec // 8, k, k), "conv"), ("fov.head.4.bias", (1,), "fov_bias"),
    ]
    return out
