"""CPU tests of the depth-model oracle (oracle/depth_pro_oracle.py).  The reference has no tests
(SURVEY §4), so the restatement is pinned by (1) op-level known answers against torch.nn.functional,
(2) the geometric invariants of split / merge / reshape_feature, (3) the committed
regression vectors tests/golden/depth_tiny_golden.npz."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import matrix_eyes_amd as m
from matrix_eyes_amd.synthetic import synthetic_checkpoint, synthetic_images
from oracle import depth_pro_oracle as O

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "depth_tiny_golden.npz")


def test_layer_norm_and_gelu_match_torch():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(7, 33, 64, generator=g) * 3 + 1
    w, b = torch.randn(64, generator=g), torch.randn(64, generator=g)
    assert torch.allclose(O.layer_norm(x, w, b, 1e-5), F.layer_norm(x, (64,), w, b, 1e-5), atol=1e-5)
    assert torch.allclose(O.gelu(x), F.gelu(x, approximate="none"), atol=1e-6)
    assert float(O.gelu(torch.tensor(1.0))) == pytest.approx(0.8413447, abs=1e-6)     # Phi(1)


def test_attention_matches_sdpa():
    cfg = O.OracleConfig(embed_dim=128, num_heads=2)
    g = torch.Generator().manual_seed(1)
    w = {"a.qkv.weight": torch.randn(384, 128, generator=g) / 11, "a.qkv.bias": torch.randn(384, generator=g) / 10,
         "a.proj.weight": torch.randn(128, 128, generator=g) / 11, "a.proj.bias": torch.randn(128, generator=g) / 10}
    x = torch.randn(3, 65, 128, generator=g)
    got = O.attention_forward(x, w, "a.", cfg)
    qkv = F.linear(x, w["a.qkv.weight"], w["a.qkv.bias"]).reshape(3, 65, 3, 2, 64).permute(2, 0, 3, 1, 4)
    ref = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2])      # scale 1/sqrt(64) (vit.rs:47)
    ref = F.linear(ref.transpose(1, 2).reshape(3, 65, 128), w["a.proj.weight"], w["a.proj.bias"])
    assert torch.allclose(got, ref, atol=2e-5)


def test_interpolate_align_corners_matches_torch():
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 3, 64, 64, generator=g)
    for out in (32, 16, 17):
        # torch evaluates the source coordinate in f32 (an error of ~4e-6 pixel), the oracle in f64
        assert torch.allclose(O.interpolate_bilinear(x, out, out, True),
                              F.interpolate(x, size=(out, out), mode="bilinear", align_corners=True), atol=1e-4)
    # corners are sampled exactly with align_corners=True
    y = O.interpolate_bilinear(x, 16, 16, True)
    assert torch.equal(y[..., 0, 0], x[..., 0, 0]) and torch.allclose(y[..., -1, -1], x[..., -1, -1])
    assert not torch.allclose(y, O.interpolate_bilinear(x, 16, 16, False), atol=1e-3)


@pytest.mark.parametrize("grid", [24, 8])
def test_split_merge_geometry(grid):
    """25 + 9 + 1 windows; merge sizes 21+18*3+21 = 96 and 18+12+18 = 48 at grid 24"""
    win = 16 * grid
    x0 = torch.arange(2 * 1 * 4 * win * 4 * win, dtype=torch.float32).reshape(2, 1, 4 * win, 4 * win)
    p0 = O.split(x0, 4, win)
    assert p0.shape == (50, 1, win, win)
    # window-major, batch-minor order (encoder.rs:149-155): window (j=1,i=2) of image 1
    stride = win - win // 4
    assert torch.equal(p0[2 * (1 * 5 + 2) + 1, 0], x0[1, 0, stride:stride + win, 2 * stride:2 * stride + win])
    p1 = O.split(x0[:, :, :2 * win, :2 * win], 2, win)
    assert p1.shape == (18, 1, win, win)
    # token-index maps through merge: every output token comes from the window that owns it
    tok = torch.arange(25 * grid * grid, dtype=torch.float32).reshape(25, 1, grid, grid)
    merged = O.merge(tok, 1, grid // 8)
    assert merged.shape == (1, 1, 4 * grid, 4 * grid)
    assert merged[0, 0, 0, 0] == tok[0, 0, 0, 0] and merged[0, 0, -1, -1] == tok[24, 0, -1, -1]
    pad = grid // 8
    first = grid - pad
    assert merged[0, 0, first, 0] == tok[5, 0, pad, 0]            # second window row starts after the crop
    m1 = O.merge(torch.zeros(9, 1, grid, grid), 1, grid // 4)
    assert m1.shape == (1, 1, 2 * grid, 2 * grid)
    if grid == 24:
        assert (grid - pad) * 2 + (grid - 2 * pad) * 3 == 96 and 18 + 12 + 18 == 48


def test_merge_of_split_reproduces_the_image():
    """merging windows of the image itself (1 'token' per pixel) gives the image back: overlapping
    crops are consistent"""
    win = 32
    img = torch.randn(1, 2, 4 * win, 4 * win)
    assert torch.equal(O.merge(O.split(img, 4, win), 1, win // 8), img)
    img1 = img[:, :, :2 * win, :2 * win]
    assert torch.equal(O.merge(O.split(img1, 2, win), 1, win // 4), img1)


def test_reshape_feature_drops_cls_and_transposes():
    emb = torch.arange(2 * 10 * 3, dtype=torch.float32).reshape(2, 10, 3)
    y = O.reshape_feature(emb, 3, 3, 1)
    assert y.shape == (2, 3, 3, 3) and torch.equal(y[1, :, 2, 1], emb[1, 1 + 2 * 3 + 1])


def test_vit_errors_like_the_reference():
    cfg = O.OracleConfig(grid=8, embed_dim=64, num_heads=1, depth=1)
    with pytest.raises(ValueError):
        O.patch_embed_forward(torch.zeros(1, 3, 130, 128), {}, "p.", cfg)       # vit.rs:213-218
    with pytest.raises(ValueError):
        O.decoder_forward([torch.zeros(1)] * 4, {}, cfg)                        # decoder.rs:161-165


def test_f_norm_from_fov_is_the_reference_formula():
    # mod.rs:358 (quirk Q1): tan(0.5 * deg * pi / 180) / 0.5
    assert O.f_norm_from_fov(90.0) == pytest.approx(2.0, rel=1e-6)
    assert O.f_norm_from_fov(53.13010235) == pytest.approx(1.0, rel=1e-5)


def test_preprocess_u8():
    rgb = np.array([[[[0, 128, 255]]]], np.uint8)
    x = O.preprocess_u8(rgb)
    assert x.shape == (1, 3, 1, 1)
    assert x.flatten().tolist() == [-1.0, float((np.float32(128) / np.float32(255) - np.float32(0.5)) / np.float32(0.5)), 1.0]


def test_oracle_reproduces_the_golden_vectors():
    gold = np.load(GOLDEN)
    cfg = m.ModelConfig.tiny()
    w = synthetic_checkpoint(cfg)
    ocfg = O.OracleConfig(grid=cfg.grid, embed_dim=cfg.embed_dim, num_heads=cfg.num_heads, depth=cfg.depth,
                          tap_blocks=cfg.tap_blocks, enc_dims=cfg.enc_dims, dec_dim=cfg.dec_dim,
                          head_dims=cfg.head_dims)
    img = O.preprocess_u8(synthetic_images(1, cfg.img_size))
    inv, fov, parts = O.extract_depth(img, None, w, ocfg, return_parts=True)
    s = int(gold["stride"])
    # threaded BLAS changes the f32 summation order, hence a tolerance instead of equality
    assert np.allclose(fov.numpy(), gold["fov_deg"], rtol=1e-5)
    assert np.allclose(parts["canonical"][0, ::s, ::s].numpy(), gold["canonical"], rtol=2e-4, atol=2e-5)
    assert np.allclose(inv[0, ::s, ::s].numpy(), gold["inverse_depth_fov"], rtol=2e-4, atol=2e-5)
    assert np.allclose(parts["lowres"][0, ::16].numpy(), gold["lowres"], rtol=2e-4, atol=2e-5)
    for i, e in enumerate(parts["encodings"]):
        st = max(1, e.shape[2] // 32)
        assert np.allclose(e[0, ::8, ::st, ::st].numpy(), gold[f"encoding{i}"], rtol=2e-4, atol=2e-5), i


def test_fp64_oracle_agrees_with_fp32():
    cfg = O.OracleConfig(grid=8, embed_dim=64, num_heads=1, depth=2, tap_blocks=(0, 1))
    mcfg = m.ModelConfig(grid=8, embed_dim=64, num_heads=1, depth=2, tap_blocks=(0, 1))
    w = synthetic_checkpoint(mcfg)
    xs = torch.randn(2, 3, 128, 128, generator=torch.Generator().manual_seed(4))
    a, _ = O.vit_forward_features(xs, w, "encoder.patch_encoder.", cfg, [])
    cfg64 = O.OracleConfig(grid=8, embed_dim=64, num_heads=1, depth=2, tap_blocks=(0, 1), dtype=torch.float64)
    b, _ = O.vit_forward_features(xs.double(), w, "encoder.patch_encoder.", cfg64, [])
    assert float((a.double() - b).abs().max()) < 1e-4


def test_oracle_wiring_against_hf_transformers():
    """The oracle is "parity unpinned" (the reference holds no fixtures and cannot be built here).  The one
    independent Depth Pro in the image is Hugging Face transformers' (written from Apple's code, not from the Rust
    reference): with ONE synthetic checkpoint converted key by key and HF put on the oracle's two assumed
    conventions (LayerNorm eps 1e-5, align_corners=True), every stage output agrees to fp32 rounding.  That is
    evidence for the wiring -- key mapping, qkv column order, window order, merge paddings, tap indices, the
    upsample / fuse / decoder / head / FOV graph -- not for the two assumed conventions themselves."""
    import importlib.util
    import json
    import os
    pytest.importorskip("transformers.models.depth_pro")
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("hf_cross_check", os.path.join(here, "hf_cross_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    before = open(os.path.join(here, "hf_cross_check.json")).read()
    try:
        mod.main()
        res = json.load(open(os.path.join(here, "hf_cross_check.json")))
    finally:
        open(os.path.join(here, "hf_cross_check.json"), "w").write(before)     # the committed fixture stays as it is
    for key in ("canonical_inverse_depth_rel_l2", "features_rel_l2", "lowres_features_rel_l2"):
        assert res[key] < 2e-5, (key, res[key])
    assert abs(res["fov_deg"][0] - res["fov_deg"][1]) < 1e-3
    committed = json.loads(before)
    assert committed["canonical_inverse_depth_rel_l2"] < 2e-5 and committed["config"] == res["config"]
