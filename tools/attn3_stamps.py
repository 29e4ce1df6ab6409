"""diagnostic: per-wave phase clocks of attention3_kernel (needs a -DME_ATT_STAMPS build of attention3.hip linked into its own
library, MATRIX_EYES_HIP_LIB pointing at it: build_ab/r05_stamps.sh)."""
import ctypes as C, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import matrix_eyes_amd as m
os.environ["ME_ATT_V"] = "3"
ctx = m.Context(0, "f16", m.ModelConfig.tiny())
lib, h = ctx.lib, ctx.handle
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
lib.me_debug_set_att3_stamps.argtypes = [C.c_void_p]
W = int(sys.argv[1]) if len(sys.argv) > 1 else 37
torch.manual_seed(1)
qkv = (torch.randn(W * 577, 3072, device="cuda") * 1.2).half()
qkv[:, :1024] = (qkv[:, :1024].float() * (0.125 * 1.4426950408889634)).half()
out = torch.empty(W * 577, 1024, dtype=torch.float16, device="cuda")
stamps = torch.zeros(4096 * 4 * 8, dtype=torch.int64, device="cuda")
NAMES = ["dma wait + barrier", "staging issue", "S = K Q^T", "max + branch", "exponentials", "P V + sums + lds wait", "whole kernel"]
p = lambda t: C.c_void_p(t.data_ptr())
FLOP = 4.0 * W * 16 * 577 * 577 * 64
launch = lambda: lib.me_op_attention_prescaled(h, p(qkv), p(out), W, 577, 16)
for it in range(3):
    stamps.zero_()
    assert lib.me_debug_set_att3_stamps(C.c_void_p(stamps.data_ptr())) == 0
    assert launch() == 0
    ctx.synchronize()
lib.me_debug_set_att3_stamps(None)
ts = []
for _ in range(10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); launch(); e1.record()
    torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
us = statistics.median(ts)
s = stamps.cpu().numpy().reshape(-1, 4, 8).astype(np.float64)
act = s[:, :, 7] > 0
a = s[act]
print(f"== attention3_kernel, {W} windows: {us:.1f} us = {FLOP / us / 1e6:.0f} TFLOP/s (stamped build, events around a launch without a stamp "
      f"buffer)   stamped workgroups {int((s[:, :, 6].sum(axis=1) > 0).sum())} active waves {int(act.sum())}", flush=True)
for i, nm in enumerate(NAMES):
    print(f"  {nm:24s} mean {a[:, i].mean():9.0f} clocks per wave  ({a[:, i].mean() / 9.0:7.0f} per tile)  share {a[:, i].sum() / a[:, 6].sum():.3f}")
print(f"  prologue (to first tile)  mean {a[:, 7].mean():9.0f} clocks per wave;  unaccounted (prologue, tail key, store stage) share {1 - a[:, :6].sum() / a[:, 6].sum():.3f}")
print(f"  (per-wave numbers are sums over the items a persistent workgroup took)")
print(f"  wave lifetime: min {a[:, 6].min():.0f}  median {np.median(a[:, 6]):.0f}  max {a[:, 6].max():.0f} clocks")
