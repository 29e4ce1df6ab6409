"""CPU tests of the host-side mirror: weight table, synthetic checkpoint, loader errors, helpers."""
import numpy as np
import pytest

import matrix_eyes_amd as m
from matrix_eyes_amd.output import _f32_round
from matrix_eyes_amd.synthetic import synthetic_checkpoint, synthetic_images


def test_expected_weights_match_the_survey():
    w = m.expected_weights(m.ModelConfig())
    names = [n for n, _, _ in w]
    assert len(names) == len(set(names)) == 1119
    total = sum(int(np.prod(s)) for _, s, _ in w)
    assert round(total / 1e6, 2) == 951.99            # 951.99 M parameters
    d = {n: s for n, s, _ in w}
    assert d["encoder.patch_encoder.pos_embed"] == (1, 577, 1024)
    assert d["encoder.patch_encoder.blocks.23.attn.qkv.weight"] == (3072, 1024)
    assert d["encoder.upsample_latent0.1.weight"] == (256, 256, 2, 2)
    assert d["encoder.fuse_lowres.weight"] == (1024, 2048, 1, 1)
    assert d["decoder.convs.4.weight"] == (256, 1024, 3, 3) and "decoder.convs.0.weight" not in d
    assert "decoder.fusions.0.deconv.weight" not in d and "decoder.fusions.1.deconv.weight" in d
    assert d["head.4.weight"] == (1, 32, 1, 1) and "head.3.weight" not in d
    assert d["fov.head.4.weight"] == (1, 32, 6, 6)
    assert d["fov.encoder.1.weight"] == (128, 1024)


def test_synthetic_checkpoint_is_seeded_and_fp16():
    import torch
    cfg = m.ModelConfig.tiny()
    a, b = synthetic_checkpoint(cfg), synthetic_checkpoint(cfg)
    assert list(a) == [n for n, _, _ in m.expected_weights(cfg)]
    assert all(t.dtype == torch.float16 for t in a.values())
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert not torch.equal(a["head.0.weight"], synthetic_checkpoint(cfg, seed=1)["head.0.weight"])
    assert float(a["head.4.bias"]) == pytest.approx(1.0, abs=0.1)


def test_synthetic_images():
    a = synthetic_images(2, 64)
    assert a.shape == (2, 64, 64, 3) and a.dtype == np.uint8
    assert np.array_equal(a, synthetic_images(2, 64)) and not np.array_equal(a[0], a[1])
    n = synthetic_images(1, 32, "noise")
    assert n.std() > 60


def test_model_config_geometry():
    c = m.ModelConfig()
    assert (c.window, c.img_size, c.tokens) == (384, 1536, 577) and m.IMG_SIZE == 1536
    t = m.ModelConfig.tiny()
    assert (t.window, t.img_size, t.tokens) == (128, 512, 65)


def test_loader_reports_a_missing_checkpoint():
    ld = m.DepthProModelLoader("/nonexistent/depth_pro.pt", False)
    with pytest.raises(m.MatrixEyesError) as e:
        ld._state_dict()
    assert e.value.code == 7 and "Failed to load depth model" in e.value.message


def test_f32_round_half_away_from_zero():
    assert _f32_round(2.5) == 3 and _f32_round(3.5) == 4 and _f32_round(0.49999997) == 0
    assert _f32_round(-1.5) == 0 and _f32_round(float("nan")) == 0     # `as u32` saturates
    assert _f32_round(296.37) == 296 and _f32_round(1e12) == 4294967295


def test_image_output_format_defaults():
    f = m.ImageOutputFormat.Stereogram(None, 1 / 16)
    assert f.kind == "stereogram" and f.resize_scale is None and f.amplitude == 0.0625
    assert m.ImageOutputFormat.DepthMap().kind == "depthmap"
    assert [int(v) for v in m.VertexMode] == [0, 1, 2]


def test_number_formatter_prints_like_rust(tmp_path):
    """csrc/ryu_f64.h -- the shortest round-trip f64 formatter that both the host writer and the device kernels of the
    OBJ writer run (output.rs:566-602 prints coordinates with Rust's `{}`) -- against std::to_chars(fixed) on 3.3 M
    doubles: random bit patterns, widened f32 values and 1 - v as the writer produces them, c / 255, powers of ten and
    two and their neighbours (tests/ryu_check.cpp states what is compared where)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "ryu_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(root, "matrix-eyes_amd", "csrc"),
                    os.path.join(root, "tests", "ryu_check.cpp"), "-o", exe], check=True, timeout=300)
    r = subprocess.run([exe, "500000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout[-2000:]
