"""Where the per-pixel tail of the full-size depth error sits: one 1536 x 1536 image through the
fp32 oracle and the f16 HIP path, then the relative error of test_extract_depth_full_size_pairs
(|d - ref| / max(|ref|, 0.05 median(ref))) against the reference value itself.
    python3 tools/tail_probe.py [family img_seed ckpt_seed]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import matrix_eyes_amd as m
from matrix_eyes_amd.synthetic import synthetic_checkpoint, synthetic_images
from oracle import depth_pro_oracle as O
from util import oracle_cfg

family = sys.argv[1] if len(sys.argv) > 1 else "structured"
img_seed = int(sys.argv[2]) if len(sys.argv) > 2 else 4321
ckpt_seed = int(sys.argv[3]) if len(sys.argv) > 3 else 2024
cfg = m.ModelConfig()
w = synthetic_checkpoint(cfg, seed=ckpt_seed)
img = O.preprocess_u8(synthetic_images(1, cfg.img_size, family, seed=img_seed))
ref, ref_fov = O.extract_depth(img, None, w, oracle_cfg(cfg))
ref = ref.numpy().astype(np.float64)
ctx = m.Context(0, "f16", cfg)
ctx.load_state_dict(w)
got, fov = ctx.extract_depth(img.numpy(), None, want_fov=True)
got = got.astype(np.float64)
med = float(np.median(ref))
err = np.abs(got - ref)
rel = err / np.maximum(np.abs(ref), 0.05 * med)
rep = {"pair": [family, img_seed, ckpt_seed], "median_ref": med, "min_ref": float(ref.min()), "max_ref": float(ref.max()),
       "rel_l2": float(np.linalg.norm(got - ref) / np.linalg.norm(ref)),
       "abs_err_over_median": {q: float(np.quantile(err, q) / med) for q in (0.5, 0.9, 0.99, 0.999, 0.9999, 1.0)},
       "ref_over_median_quantiles": {q: float(np.quantile(ref, q) / med) for q in (0.001, 0.01, 0.05, 0.25, 0.5, 0.75, 0.99)}}
for thr in (2e-3, 5e-3, 1e-2, 5e-2):
    bad = rel > thr
    rep[f"rel>{thr:g}"] = {
        "fraction": float(bad.mean()),
        "ref_over_median_of_those": ({q: float(np.quantile(ref[bad], q) / med) for q in (0.5, 0.9, 0.99, 1.0)} if bad.any() else None),
        "abs_err_over_median_of_those": ({q: float(np.quantile(err[bad], q) / med) for q in (0.5, 0.99, 1.0)} if bad.any() else None),
    }
# relative error by band of the reference value
bands = [0, 0.02, 0.05, 0.1, 0.2, 0.5, 1.0, 2.0, 1e9]
rep["by_ref_band"] = []
for lo, hi in zip(bands, bands[1:]):
    sel = (ref >= lo * med) & (ref < hi * med)
    if sel.any():
        rep["by_ref_band"].append({"ref/median": [lo, hi], "pixels": int(sel.sum()),
                                   "rel_median": float(np.median(rel[sel])), "rel_p99": float(np.quantile(rel[sel], 0.99)),
                                   "rel_max": float(rel[sel].max()), "abs_over_median_max": float(err[sel].max() / med),
                                   "abs_over_median_p99": float(np.quantile(err[sel], 0.99) / med)})
print(json.dumps(rep, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rep, open(f"gpurun_out/tail_probe_{family}_{img_seed}_{ckpt_seed}.json", "w"), indent=1)
