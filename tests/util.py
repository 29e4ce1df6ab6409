"""Shared helpers of the GPU parity tests: device tensors come from torch (plumbing only), every
compute call goes through the C ABI of libmatrixeyes_hip.so."""
import ctypes as C

import numpy as np
import torch

import matrix_eyes_amd as m
from oracle import depth_pro_oracle as O

TORCH16 = {"f16": torch.float16, "bf16": torch.bfloat16}
_CTX = {}


def ctx_for(cfg_name: str, dtype: str):
    """One context per (config, dtype) for the whole session (weights loaded once)."""
    key = (cfg_name, dtype)
    if key not in _CTX:
        cfg = {"tiny": m.ModelConfig.tiny(), "full": m.ModelConfig()}[cfg_name]
        ctx = m.Context(0, dtype, cfg)
        _CTX[key] = ctx
    return _CTX[key]


_WEIGHTS = {}


def weights_for(cfg_name: str):
    from matrix_eyes_amd.synthetic import synthetic_checkpoint
    if cfg_name not in _WEIGHTS:
        cfg = {"tiny": m.ModelConfig.tiny(), "full": m.ModelConfig()}[cfg_name]
        _WEIGHTS[cfg_name] = synthetic_checkpoint(cfg)
    return _WEIGHTS[cfg_name]


def loaded_ctx(cfg_name: str, dtype: str):
    key = (cfg_name, dtype, "loaded")
    if key not in _CTX:
        ctx = ctx_for(cfg_name, dtype)
        ctx.load_state_dict(weights_for(cfg_name))
        _CTX[key] = ctx
    return _CTX[key]


def oracle_cfg(cfg: m.ModelConfig, dtype=torch.float32) -> O.OracleConfig:
    return O.OracleConfig(grid=cfg.grid, embed_dim=cfg.embed_dim, num_heads=cfg.num_heads,
                          depth=cfg.depth, tap_blocks=tuple(cfg.tap_blocks), enc_dims=tuple(cfg.enc_dims),
                          dec_dim=cfg.dec_dim, head_dims=tuple(cfg.head_dims), ln_eps=cfg.ln_eps,
                          align_corners=cfg.align_corners, dtype=dtype)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def dev16(x: torch.Tensor, dtype: str) -> torch.Tensor:
    return x.to(TORCH16[dtype]).cuda().contiguous()


def rel_l2(a, b) -> float:
    a = torch.as_tensor(a, dtype=torch.float64).flatten()
    b = torch.as_tensor(b, dtype=torch.float64).flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_abs_rel(a, b) -> float:
    """max |a-b| / rms(b): worst element against the tensor's own scale"""
    a = torch.as_tensor(a, dtype=torch.float64).flatten()
    b = torch.as_tensor(b, dtype=torch.float64).flatten()
    return float((a - b).abs().max() / b.pow(2).mean().sqrt().clamp_min(1e-30))


def max_err_over_max(a, b) -> float:
    """max |a-b| / max |b|: worst element against the largest magnitude (what one rounding of a
    16-bit output is proportional to)"""
    a = torch.as_tensor(a, dtype=torch.float64).flatten()
    b = torch.as_tensor(b, dtype=torch.float64).flatten()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def pack_conv(w: torch.Tensor) -> torch.Tensor:
    """[Cout][Cin][kh][kw] -> [Cout][kh*kw][Cin] (csrc/weights.hip PK_CONV_16)"""
    co, ci, kh, kw = w.shape
    return w.permute(0, 2, 3, 1).reshape(co, kh * kw * ci).contiguous()


def pack_convt(w: torch.Tensor) -> torch.Tensor:
    """[Cin][Cout][2][2] -> [(dy*2+dx)*Cout + co][Cin] (PK_CONVT_16)"""
    ci, co, _, _ = w.shape
    return w.permute(2, 3, 1, 0).reshape(4 * co, ci).contiguous()


def bordered(x_nchw: torch.Tensor, dtype: str) -> torch.Tensor:
    """NCHW f32 -> zero-bordered NHWC 16-bit [B][H+2][W+2][C] on the device"""
    b, c, h, w = x_nchw.shape
    out = torch.zeros((b, h + 2, w + 2, c), dtype=TORCH16[dtype], device="cuda")
    out[:, 1:h + 1, 1:w + 1, :] = x_nchw.permute(0, 2, 3, 1).to(TORCH16[dtype]).cuda()
    return out.contiguous()


def depth_error_report(d, ref):
    """Error figures of an inverse-depth map against the oracle's (both [B,S,S] f32 arrays).
    rel = |d - ref| / max(|ref|, 0.05 * median(ref)): relative error with a floor so that the pixels
    the closing ReLU zeroes (clamped to 1e-4 on both sides) do not divide by ~0.
    The per-pixel tail (tools/tail_probe.py): the error is ADDITIVE -- |d - ref| has the same distribution in every
    band of the reference value (16-bit operand roundings upstream of the head leave a noise floor proportional to the
    map's own scale, not to the pixel's value) -- so it is measured against the map's rms (abs_*), and the pixels whose
    RELATIVE error is large are the ones whose reference value is small against that scale (tail_*)."""
    d = np.asarray(d, np.float64)
    ref = np.asarray(ref, np.float64)
    med = float(np.median(ref))
    floor = 0.05 * med
    err = np.abs(d - ref)
    rel = err / np.maximum(np.abs(ref), floor)
    rms = float(np.sqrt(np.mean(ref * ref)))
    bad = rel > 1e-2
    bright = ref >= rms
    return {
        "rel_l2": float(np.linalg.norm(d - ref) / np.linalg.norm(ref)),
        "median": float(np.median(rel)),
        "p99": float(np.quantile(rel, 0.99)),
        "max": float(rel.max()),
        "rms_over_median": rms / med,
        "abs_over_rms_median": float(np.median(err) / rms),
        "abs_over_rms_p99": float(np.quantile(err, 0.99) / rms),
        "abs_over_rms_max": float(err.max() / rms),
        "tail_fraction": float(bad.mean()),                                      # pixels with rel > 1e-2
        "tail_ref_over_rms_max": float(ref[bad].max() / rms) if bad.any() else 0.0,
        "bright_fraction": float(bright.mean()),                                 # pixels with ref >= rms(ref)
        "bright_rel_max": float(rel[bright].max()) if bright.any() else 0.0,
    }
