"""diagnostic: per-wave phase times of the attention kernel (needs a -DME_ATT_STAMPS build of the library, e.g.
make -C matrix-eyes_amd/csrc BUILD=../build_stamps OUT=../../build_ab/libstamps.so CXXFLAGS="... -DME_ATT_STAMPS";
run with MATRIX_EYES_HIP_LIB pointing at it)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import matrix_eyes_amd as m
from tools.bench_kernels import ptr
ctx = m.Context(0, "f16", m.ModelConfig.tiny())
lib, h = ctx.lib, ctx.handle
lib.me_debug_set_att_stamps.argtypes = [C.c_void_p]
W = 35
qkv = torch.randn(W * 577, 3072, device="cuda").half()
out = torch.empty(W * 577, 1024, dtype=torch.float16, device="cuda")
stamps = torch.zeros(4096 * 4 * 8, dtype=torch.int64, device="cuda")
for it in range(3):
    stamps.zero_()
    assert lib.me_debug_set_att_stamps(C.c_void_p(stamps.data_ptr())) == 0
    lib.me_op_attention(h, ptr(qkv), ptr(out), W, 577, 16)
    ctx.synchronize()
lib.me_debug_set_att_stamps(None)
s = stamps.cpu().numpy().reshape(-1, 4, 8).astype(np.float64)
act = s[:, :, 7] > 0
names = ["dma wait + barrier", "staging issue", "S mfma + max", "rescale + exp", "P V", "closing lds wait", "whole kernel"]
print("workgroups", int((s[:, :, 6].sum(axis=1) > 0).sum()), "active waves", int(act.sum()))
a = s[act]
tiles = 9.0
for i, nm in enumerate(names):
    print(f"  {nm:20s} mean {a[:, i].mean():9.0f} clocks per wave  ({a[:, i].mean() / tiles:7.0f} per tile)  share {a[:, i].sum() / a[:, 6].sum():.3f}")
print("  unaccounted (prologue, tail key, epilogue) share %.3f" % (1 - a[:, :6].sum() / a[:, 6].sum()))
