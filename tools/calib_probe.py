"""What on a box predicts its step time?  (the MFMA-only calibration loop does not, to better than +-2 %.)
On ONE box: the two fixed loops cold and after the step has run for a second; the MFMA loop with HBM copies running beside it
on another stream; a long run of the qkv-shaped GEMM probe; the step itself.  Run on several boxes and compare.
    python3 tools/calib_probe.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import matrix_eyes_amd as m
from matrix_eyes_amd.synthetic import synthetic_checkpoint, synthetic_images

cfg = m.ModelConfig()
ctx = m.Context(0, "f16", cfg)
ctx.load_state_dict(synthetic_checkpoint(cfg))
rgb = torch.from_numpy(synthetic_images(1, cfg.img_size, "structured", seed=4321)).cuda()
out = torch.empty(1, cfg.img_size, cfg.img_size, dtype=torch.float32, device="cuda")
res = {}
res["cold"] = ctx.calibrate()


def step_ms(n):
    for _ in range(3):
        ctx.extract_depth(rgb, None, out=out)
    ctx.synchronize()
    t0 = time.time()
    for _ in range(n):
        ctx.extract_depth(rgb, None, out=out)
    ctx.synchronize()
    return (time.time() - t0) / n * 1e3


res["step_ms_20"] = step_ms(20)
res["after_20_steps"] = ctx.calibrate()
res["step_ms_100"] = step_ms(100)
res["after_100_steps"] = ctx.calibrate()
# the MFMA loop beside HBM traffic: 400 x 512 MiB device copies queued on another stream first (about 90 ms)
a = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
b = torch.empty_like(a)
side = torch.cuda.Stream()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(side):
    e0.record()
    for _ in range(400):
        b.copy_(a, non_blocking=True)
    e1.record()
res["mfma_beside_copies"] = ctx.calibrate()
torch.cuda.synchronize()
res["copies_beside_mfma_gbs"] = 400 * 2 * (512 << 20) / (e0.elapsed_time(e1) * 1e-3) / 1e9
res["step_ms_20_again"] = step_ms(20)
print(json.dumps({k: (v if not isinstance(v, dict) else {a_: round(b_, 3) for a_, b_ in v.items()}) for k, v in res.items()}))
