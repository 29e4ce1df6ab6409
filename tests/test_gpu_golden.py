"""-m gpu: the HIP path against the committed regression vectors (tests/golden/depth_tiny_golden.npz,
made by tests/golden/make_golden.py from the CPU oracle) — no CPU forward pass needed."""
import os

import numpy as np
import pytest

from matrix_eyes_amd.synthetic import synthetic_images
from util import loaded_ctx, rel_l2

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "depth_tiny_golden.npz")


@pytest.mark.parametrize("dtype,tol", [("f16", 2e-3), ("bf16", 1.6e-2)])
def test_against_golden(dtype, tol):
    gold = np.load(GOLDEN)
    ctx = loaded_ctx("tiny", dtype)
    rgb = synthetic_images(1, ctx.cfg.img_size)
    s = int(gold["stride"])
    d, fov = ctx.extract_depth(rgb, None, want_fov=True)
    assert rel_l2(d[0, ::s, ::s], gold["inverse_depth_fov"]) < tol
    assert abs(float(fov[0]) - float(gold["fov_deg"][0])) < (0.05 if dtype == "f16" else 0.4)
    d1 = ctx.extract_depth(rgb, 1.0)
    assert rel_l2(d1[0, ::s, ::s], gold["inverse_depth_fnorm1"]) < tol
    img = ctx.preprocess_u8(rgb)
    enc = ctx.encoder_forward_encodings(img)
    for i, e in enumerate(enc):
        st = max(1, e.shape[2] // 32)
        assert rel_l2(e[0, ::8, ::st, ::st], gold[f"encoding{i}"]) < tol, i
