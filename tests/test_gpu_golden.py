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


def test_bench_line_contract():
    """bench.py prints ONE JSON line with the driver's keys, a roofline object measured live and a CPU baseline."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-windows", "1"],
                       capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f16"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 0.02          # one image per step
    rf = d["roofline"]
    assert rf["bound"] in ("mfma", "hbm") and rf["unit"] in ("TFLOP/s", "GB/s")
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0.05 < rf["frac"] < 1.0
    assert rf["traffic"] is None or rf["traffic"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert d["value"] > 100 * cb["value"]
