"""configs[4] OBJ leg, leg by leg (ME_OBJ_TIMING=1 prints mesh / format / D2H / file times from inside me_output_mesh):
one full-size depth map, textured OBJ on tmpfs, device formatter and host formatter."""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ME_OBJ_TIMING"] = "1"
import numpy as np, torch
import matrix_eyes_amd as m
ctx = m.Context(0, "f16", m.ModelConfig.tiny())
S = 1536
rng = np.random.default_rng(0)
yy, xx = np.meshgrid(np.linspace(0, 1, S, dtype=np.float32), np.linspace(0, 1, S, dtype=np.float32), indexing="ij")
for name, d in (("smooth (every face kept)", 0.2 + 0.15 * np.sin(3 * xx + 2 * yy) + 0.1 * yy),
                ("rough (random-weight-like)", np.exp(rng.normal(0, 0.03, size=(S, S))).astype(np.float32) * 0.3)):
    depth = torch.from_numpy(np.ascontiguousarray(d.astype(np.float32))).cuda()
    ddm = m.DeviceDepthMap(ctx, depth, (S, S))
    out = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    for i in range(4):
        t0 = time.perf_counter()
        ddm.output_mesh(os.path.join(out, "mesh.obj"), "photo.jpg", m.VertexMode.Texture)
        print(name, "call", i, f"{(time.perf_counter() - t0) * 1e3:.1f} ms", os.path.getsize(os.path.join(out, "mesh.obj")), flush=True)
    shutil.rmtree(out)
