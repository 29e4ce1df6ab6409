"""Seeded synthetic checkpoint and images: the real depth_pro.pt (1.9 GB, fp16) is
not available offline, so benchmarks and parity tests run on random-init weights with the exact
key set and shapes, drawn in f32 and rounded to fp16 like the real file."""
from typing import Dict

import numpy as np
import torch

from .config import ModelConfig, expected_weights


def _fan_in(shape, kind):
    if kind == "linear":
        return shape[1]
    if kind == "conv":
        return shape[1] * shape[2] * shape[3]
    if kind == "convt":   # stride == kernel: every output pixel sees Cin inputs through one tap
        return shape[0]
    raise ValueError(kind)


def synthetic_checkpoint(cfg: ModelConfig, seed: int = 2024) -> Dict[str, torch.Tensor]:
    """name -> fp16 CPU tensor in the PyTorch layout."""
    gen = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape, kind in expected_weights(cfg):
        if kind in ("linear", "conv", "convt"):
            t = torch.randn(shape, generator=gen) * (1.0 / _fan_in(shape, kind)) ** 0.5
        elif kind == "bias":
            t = torch.randn(shape, generator=gen) * 0.02
        elif kind == "head_bias":       # keeps the closing ReLU mostly active (output.rs:129,178
            t = torch.randn(shape, generator=gen) * 0.02 + 1.0   # divide by max - min)
        elif kind == "fov_bias":        # a plausible field of view, degrees
            t = torch.randn(shape, generator=gen) * 0.02 + 55.0
        elif kind == "ln_weight":
            t = 1.0 + torch.randn(shape, generator=gen) * 0.02
        elif kind == "ln_bias":
            t = torch.randn(shape, generator=gen) * 0.02
        elif kind == "layer_scale":
            t = 0.05 + 0.15 * torch.rand(shape, generator=gen)
        elif kind == "embed":
            t = torch.randn(shape, generator=gen) * 0.02
        else:
            raise ValueError(kind)
        out[name] = t.to(torch.float16)
    return out


def synthetic_images(batch: int, size: int, family: str = "structured", seed=None) -> np.ndarray:
    """u8 [batch, size, size, 3].  'noise': uniform bytes (seed 1234); 'structured': low-frequency
    sinusoids + rectangles (seed 4321), which gives a non-degenerate depth map."""
    if family == "noise":
        rng = np.random.default_rng(1234 if seed is None else seed)
        return rng.integers(0, 256, size=(batch, size, size, 3), dtype=np.uint8)
    rng = np.random.default_rng(4321 if seed is None else seed)
    yy, xx = np.meshgrid(np.arange(size, dtype=np.float32), np.arange(size, dtype=np.float32),
                         indexing="ij")
    out = np.empty((batch, size, size, 3), dtype=np.uint8)
    for b in range(batch):
        img = np.zeros((size, size, 3), dtype=np.float32)
        for _ in range(6):
            fx, fy = rng.uniform(0.5, 4.0, size=2) * 2 * np.pi / size
            ph = rng.uniform(0, 2 * np.pi, size=3)
            amp = rng.uniform(0.1, 0.35)
            for c in range(3):
                img[..., c] += amp * np.sin(fx * xx + fy * yy + ph[c])
        img = 0.5 + 0.5 * img / max(1e-6, np.abs(img).max())
        for _ in range(8):
            x0, y0 = rng.integers(0, size - size // 8, size=2)
            w, h = rng.integers(size // 16, size // 4, size=2)
            img[y0:y0 + h, x0:x0 + w] = rng.uniform(0, 1, size=3)
        out[b] = np.clip(np.round(img * 255.0), 0, 255).astype(np.uint8)
    return out
