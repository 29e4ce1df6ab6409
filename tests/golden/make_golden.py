"""Generates tests/golden/depth_tiny_golden.npz: outputs of the CPU oracle
(oracle/depth_pro_oracle.py, fp32) on the tiny model configuration with the seeded synthetic
checkpoint (seed 2024) and the 'structured' synthetic image (seed 4321).

The reference itself cannot produce vectors here (Rust on un-vendored crates, no toolchain: SURVEY
§8c), so these are REGRESSION vectors of the restatement, not reference outputs: they pin the oracle
against accidental change (tests/test_oracle_depth_pro.py) and give the GPU tests a target that needs
no CPU forward (tests/test_gpu_golden.py).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import matrix_eyes_amd as m  # noqa: E402
from matrix_eyes_amd.synthetic import synthetic_checkpoint, synthetic_images  # noqa: E402
from oracle import depth_pro_oracle as O  # noqa: E402

STRIDE = 8


def main():
    torch.set_num_threads(1)     # one summation order, whatever machine regenerates the file
    cfg = m.ModelConfig.tiny()
    w = synthetic_checkpoint(cfg)
    ocfg = O.OracleConfig(grid=cfg.grid, embed_dim=cfg.embed_dim, num_heads=cfg.num_heads, depth=cfg.depth,
                          tap_blocks=cfg.tap_blocks, enc_dims=cfg.enc_dims, dec_dim=cfg.dec_dim,
                          head_dims=cfg.head_dims)
    img = O.preprocess_u8(synthetic_images(1, cfg.img_size))
    inv_fov, fov, parts = O.extract_depth(img, None, w, ocfg, return_parts=True)
    inv_one, _ = O.extract_depth(img, 1.0, w, ocfg)
    out = {
        "stride": np.int32(STRIDE),
        "fov_deg": fov.numpy().astype(np.float32),
        "inverse_depth_fov": inv_fov[0, ::STRIDE, ::STRIDE].numpy(),
        "inverse_depth_fnorm1": inv_one[0, ::STRIDE, ::STRIDE].numpy(),
        "canonical": parts["canonical"][0, ::STRIDE, ::STRIDE].numpy(),
        "features": parts["features"][0, ::32, ::STRIDE, ::STRIDE].numpy(),
        "lowres": parts["lowres"][0, ::16].numpy(),
    }
    for i, e in enumerate(parts["encodings"]):
        s = max(1, e.shape[2] // 32)
        out[f"encoding{i}"] = e[0, ::8, ::s, ::s].numpy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "depth_tiny_golden.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
