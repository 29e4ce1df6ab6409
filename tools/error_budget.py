"""diagnostic: where the f16 path's error against the fp32 oracle comes from, full-size model.
Each stage is fed the ORACLE's inputs, so the figures are per-stage, not accumulated."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import matrix_eyes_amd as m
from matrix_eyes_amd.synthetic import synthetic_checkpoint, synthetic_images
from oracle import depth_pro_oracle as O
from util import oracle_cfg, rel_l2, depth_error_report

cfg = m.ModelConfig() if (len(sys.argv) < 2 or sys.argv[1] == "full") else m.ModelConfig.tiny()
dtype = sys.argv[2] if len(sys.argv) > 2 else "f16"
w = synthetic_checkpoint(cfg)
ctx = m.Context(0, dtype, cfg); ctx.load_state_dict(w)
img = O.preprocess_u8(synthetic_images(1, cfg.img_size))
t = time.time(); inv, fov, parts = O.extract_depth(img, None, w, oracle_cfg(cfg), return_parts=True); print("oracle s", time.time() - t, flush=True)
enc = ctx.encoder_forward_encodings(img.numpy())
for i, (g, r) in enumerate(zip(enc, parts["encodings"])): print("encoder out", i, "rel_l2", rel_l2(g, r), flush=True)
feat, low = ctx.decoder_forward([e.numpy() for e in parts["encodings"]])
print("decoder (oracle encodings in): features", rel_l2(feat, parts["features"]), "lowres", rel_l2(low, parts["lowres"]))
feat2, _ = ctx.decoder_forward(enc)
print("decoder (gpu encodings in): features", rel_l2(feat2, parts["features"]))
canon = ctx.head_forward(parts["features"].numpy())
print("head (oracle features in): canonical", rel_l2(canon, parts["canonical"]))
canon2 = ctx.head_forward(feat2)
print("head (gpu features in): canonical", rel_l2(canon2, parts["canonical"]))
d, f = ctx.extract_depth(img.numpy(), None, want_fov=True)
print("end to end", depth_error_report(d, inv.numpy()), float(f[0]), float(fov[0]))
# ViT alone on 2 windows, with taps
xs = img[:, :, :cfg.window, :2 * cfg.window].reshape(1, 3, cfg.window, 2, cfg.window).permute(0, 3, 1, 2, 4).reshape(2, 3, cfg.window, cfg.window).contiguous()
fin, inter = ctx.vit_forward_features(0, xs.numpy(), list(cfg.tap_blocks))
rf, ri = O.vit_forward_features(xs, w, "encoder.patch_encoder.", oracle_cfg(cfg), list(cfg.tap_blocks))
print("vit final", rel_l2(fin, rf), "taps", rel_l2(inter[0], ri[0]), rel_l2(inter[1], ri[1]))
