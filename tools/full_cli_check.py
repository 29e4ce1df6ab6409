"""Manual full-size check of the C++ host layer: writes the 1.9 GB synthetic depth_pro.pt to /tmp, runs
matrix-eyes-hip on a 1536x1536 PNG and compares its depth map with the Python mirror's (identical pixels
expected).  Too heavy for the test suite; tests/test_host_cpp.py does the same on the test geometry."""
import os, subprocess, sys, time, numpy as np, torch
sys.path.insert(0, ".")
import matrix_eyes_amd as m
from matrix_eyes_amd.synthetic import synthetic_checkpoint, synthetic_images
from matrix_eyes_amd import reconstruction as R
from PIL import Image
cfg = m.ModelConfig()
d = "/tmp/fullcli"; os.makedirs(d, exist_ok=True)
t = time.time()
ck = {k: torch.as_tensor(v) for k, v in synthetic_checkpoint(cfg).items()}
torch.save(ck, d + "/depth_pro.pt"); print("checkpoint written", round(time.time() - t, 1), "s", os.path.getsize(d + "/depth_pro.pt") / 1e9, "GB")
Image.fromarray(synthetic_images(1, 1536, "structured", seed=3)[0]).save(d + "/photo.png")
t = time.time()
r = subprocess.run(["matrix-eyes_amd/matrix-eyes-hip", f"--checkpoint-path={d}/depth_pro.pt", d + "/photo.png", d + "/depth_cpp.png"], capture_output=True, text=True)
print("cli rc", r.returncode, round(time.time() - t, 1), "s", r.stdout.strip()[-200:], r.stderr.strip()[-200:])
loader = m.DepthProModelLoader(d + "/depth_pro.pt", False, cfg)
R.extract_depth(0, loader, d + "/photo.png", d + "/depth_py.png", None, m.ImageOutputFormat.DepthMap(), m.VertexMode.Color)
a, b = np.asarray(Image.open(d + "/depth_cpp.png")), np.asarray(Image.open(d + "/depth_py.png"))
print("full-size depth map identical:", np.array_equal(a, b), a.shape)
