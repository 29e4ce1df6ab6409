"""attention3_kernel's persistent grid (ME_ATT_GRID, development): time at the step's shape for several grids -- the resident count
(default), one workgroup per item (1776: the non-persistent order with this code), and points between."""
import ctypes as C, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import matrix_eyes_amd as m
os.environ["ME_ATT_V"] = "3"
ctx = m.Context(0, "f16", m.ModelConfig.tiny())
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
W = int(sys.argv[1]) if len(sys.argv) > 1 else 37
torch.manual_seed(1)
qkv = (torch.randn(W * 577, 3072, device="cuda") * 1.2).half()
qkv[:, :1024] = (qkv[:, :1024].float() * (0.125 * 1.4426950408889634)).half()
out = torch.empty(W * 577, 1024, dtype=torch.float16, device="cuda")
p = lambda t: C.c_void_p(t.data_ptr())
FLOP = 4.0 * W * 16 * 577 * 577 * 64
grids = [0, 256, 384, 512, 592, 640, 768, 888, 1024, 1184, 1776]
ts = {g: [] for g in grids}
for r in range(8):
    for g in grids:
        if g:
            os.environ["ME_ATT_GRID"] = str(g)
        else:
            os.environ.pop("ME_ATT_GRID", None)
        ctx.lib.me_op_attention_prescaled(ctx.handle, p(qkv), p(out), W, 577, 16)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            ctx.lib.me_op_attention_prescaled(ctx.handle, p(qkv), p(out), W, 577, 16)
        e1.record(); torch.cuda.synchronize()
        ts[g].append(e0.elapsed_time(e1) * 1e3 / 4)
for g in grids:
    med = statistics.median(ts[g])
    print(f"grid {g if g else 'resident (default)':>20}: median {med:7.1f} us  min {min(ts[g]):7.1f} us  {FLOP / med / 1e6:5.0f} TFLOP/s", flush=True)
