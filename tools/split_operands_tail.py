"""The per-pixel error figures of the full-size depth map for each `split_operands` mask ("split_operands=15
buys 5.2e-4 for +1.7 ms; nobody has shown what the per-pixel figures are there"): ONE 1536 x 1536 image through the fp32 oracle,
then the f16 HIP path with masks 0, 3 (the default), 7, 15 -- tests/util.py depth_error_report on each and the step time.
    python3 tools/split_operands_tail.py [mask ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import dataclasses
import numpy as np
import torch
import matrix_eyes_amd as m
from matrix_eyes_amd.synthetic import synthetic_checkpoint, synthetic_images
from oracle import depth_pro_oracle as O
from util import depth_error_report, oracle_cfg

masks = [int(a) for a in sys.argv[1:]] or [0, 3, 7, 15]
base = m.ModelConfig()
w = synthetic_checkpoint(base, seed=2024)
rgb = synthetic_images(1, base.img_size, "structured", seed=4321)
t0 = time.time()
ref, ref_fov = O.extract_depth(O.preprocess_u8(rgb), None, w, oracle_cfg(base))
print(f"oracle {time.time() - t0:.1f} s", flush=True)
ref = ref.numpy()
dev = torch.from_numpy(rgb).cuda()
out = torch.empty(1, base.img_size, base.img_size, dtype=torch.float32, device="cuda")
for mask in masks:
    cfg = dataclasses.replace(base, split_operands=mask)
    ctx = m.Context(0, "f16", cfg)
    ctx.load_state_dict(w)
    d, fov = ctx.extract_depth(rgb, None, want_fov=True)
    rep = depth_error_report(d, ref)
    for _ in range(3):
        ctx.extract_depth(dev, None, out=out)
    ctx.synchronize()
    t0 = time.time()
    for _ in range(10):
        ctx.extract_depth(dev, None, out=out)
    ctx.synchronize()
    ms = (time.time() - t0) / 10 * 1e3
    print(f"split_operands {mask:2d}: rel_l2 {rep['rel_l2']:.3e}  rel median {rep['median']:.2e} p99 {rep['p99']:.2e} max {rep['max']:.2e}  "
          f"|d-ref|/rms median {rep['abs_over_rms_median']:.2e} p99 {rep['abs_over_rms_p99']:.2e} max {rep['abs_over_rms_max']:.2e}  "
          f"pixels >= rms: rel max {rep['bright_rel_max']:.2e}  fov {float(fov[0]):.4f} vs {float(ref_fov[0]):.4f}  {ms:.2f} ms/step", flush=True)
    ctx.close()
