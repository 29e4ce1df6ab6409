"""BASELINE configs[3] accuracy / throughput budget (GPU box): the full-size model with each subset of the four ViT
linears on MX fp8 (me_model_config.fp8_linears: 1 = qkv, 2 = proj, 4 = fc1, 8 = fc2; the rest on the f16 kernels),
depth error against the fp32 oracle and time per step for each mask.  python tools/fp8_budget.py [mask ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import matrix_eyes_amd as m
from matrix_eyes_amd.synthetic import synthetic_checkpoint, synthetic_images
from oracle import depth_pro_oracle as O
from util import oracle_cfg, depth_error_report

masks = [int(x) for x in sys.argv[1:]] or [-1, 1, 2, 4, 8, 3, 12, 5, 13, 14, 15]
base = m.ModelConfig()
w = synthetic_checkpoint(base)
rgb = synthetic_images(1, base.img_size)
img = O.preprocess_u8(rgb)
t = time.time()
inv, fov = O.extract_depth(img, None, w, oracle_cfg(base))
print(f"oracle {time.time() - t:.1f} s", flush=True)
names = {1: "qkv", 2: "proj", 4: "fc1", 8: "fc2"}
for mask in masks:
    if mask < 0:
        ctx, label = m.Context(0, "f16", base), "f16 (no fp8)"
    else:
        ctx = m.Context(0, "fp8", m.ModelConfig(**{**base.__dict__, "fp8_linears": mask}))
        label = "+".join(n for b, n in names.items() if mask & b)
    ctx.load_state_dict(w)
    d, f = ctx.extract_depth(img.numpy(), None, want_fov=True)
    dev = torch.from_numpy(rgb).cuda()
    out = torch.empty(1, base.img_size, base.img_size, dtype=torch.float32, device="cuda")
    for _ in range(3): ctx.extract_depth(dev, None, out=out)
    ctx.synchronize(); t = time.time()
    for _ in range(10): ctx.extract_depth(dev, None, out=out)
    ctx.synchronize(); ms = (time.time() - t) * 100
    rep = depth_error_report(d, inv.numpy())
    print(f"mask {mask:2d} {label:18s}: rel_l2 {rep['rel_l2']:.3e} median {rep['median']:.2e} p99 {rep['p99']:.2e} "
          f"fov {float(f[0]):.4f} vs {float(fov[0]):.4f}  {ms:.2f} ms/step = {1000 / ms:.1f} depth-maps/s", flush=True)
    ctx.close()
    del ctx
    torch.cuda.empty_cache()
