"""diagnostic (GPU box): depth error of the full-size f16 path against the fp32 oracle for several split-operand
masks (csrc/model.h SplitStage: 1 upsample chains, 2 fusion deconv/out_conv, 4 head, 8 decoder.convs), with the
step time beside each.  python tools/split_budget.py [full|tiny] [dtype] [mask ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import matrix_eyes_amd as m
from matrix_eyes_amd.synthetic import synthetic_checkpoint, synthetic_images
from oracle import depth_pro_oracle as O
from util import oracle_cfg, depth_error_report

size = sys.argv[1] if len(sys.argv) > 1 else "full"
dtype = sys.argv[2] if len(sys.argv) > 2 else "f16"
masks = [int(x) for x in sys.argv[3:]] or [0, 1, 2, 4, 7, 15]
cfg = m.ModelConfig() if size == "full" else m.ModelConfig.tiny()
w = synthetic_checkpoint(cfg)
rgb = synthetic_images(1, cfg.img_size)
img = O.preprocess_u8(rgb)
t = time.time()
inv, fov = O.extract_depth(img, None, w, oracle_cfg(cfg))
print(f"oracle {time.time() - t:.1f} s", flush=True)
for mask in masks:
    cfg = m.ModelConfig(**{**cfg.__dict__, "split_operands": mask})
    ctx = m.Context(0, dtype, cfg)
    ctx.load_state_dict(w)
    d, f = ctx.extract_depth(img.numpy(), None, want_fov=True)
    dev = torch.from_numpy(rgb).cuda()
    out = torch.empty(1, cfg.img_size, cfg.img_size, dtype=torch.float32, device="cuda")
    for _ in range(3): ctx.extract_depth(dev, None, out=out)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(10): ctx.extract_depth(dev, None, out=out)
    ctx.synchronize(); torch.cuda.synchronize(); ms = (time.time() - t) * 100
    rep = depth_error_report(d, inv.numpy())
    print(f"mask {mask:2d}: rel_l2 {rep['rel_l2']:.3e} median {rep['median']:.2e} p99 {rep['p99']:.2e} "
          f"fov {float(f[0]):.5f} vs {float(fov[0]):.5f}  {ms:.2f} ms/step  arena {ctx.weight_arena_bytes() / 1e9:.3f} GB", flush=True)
    ctx.close()
    del ctx
    torch.cuda.empty_cache()
