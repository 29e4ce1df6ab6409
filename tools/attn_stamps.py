"""diagnostic: per-wave phase times of the attention kernels (needs a -DME_ATT_STAMPS build of the library:
make -C matrix-eyes_amd/csrc BUILD=../build_stamps OUT=../../build_ab/libstamps.so CXXFLAGS="... -DME_ATT_STAMPS";
run with MATRIX_EYES_HIP_LIB pointing at it).  attention2_kernel also takes timing-only ablations (results wrong):
1 = no restaging / barrier after the first tile, 2 = no exponentials, 4 = no P V MFMAs, 8 = no S MFMAs."""
import ctypes as C, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import matrix_eyes_amd as m
from tools.bench_kernels import ptr
ctx = m.Context(0, "f16", m.ModelConfig.tiny())
lib, h = ctx.lib, ctx.handle
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)   # events and launches on one stream
lib.me_debug_set_att_stamps.argtypes = [C.c_void_p]
lib.me_debug_set_att_mode.argtypes = [C.c_int32]
W = 37
qkv = (torch.randn(W * 577, 3072, device="cuda") * 1.2).half()
out = torch.empty(W * 577, 1024, dtype=torch.float16, device="cuda")
stamps = torch.zeros(4096 * 4 * 8, dtype=torch.int64, device="cuda")
NAMES = {"1": ["dma wait + barrier", "staging issue", "S mfma + max", "rescale + exp", "P V", "closing lds wait", "whole kernel"],
         "2": ["dma wait + barrier", "staging issue", "S = K Q^T", "max + branch", "exponentials", "P V + sums + lds wait", "whole kernel"]}


def timed(n=8):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream()); lib.me_op_attention(h, ptr(qkv), ptr(out), W, 577, 16); e1.record(torch.cuda.current_stream())
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)


for ver, minw, mode in (("1", "", 0), ("2", "3", 0), ("2", "3", 1), ("2", "3", 2), ("2", "3", 4), ("2", "3", 8)):
    os.environ["ME_ATT_V"] = ver
    if minw:
        os.environ["ME_ATT_MINW"] = minw
    assert lib.me_debug_set_att_mode(mode) == 0
    for it in range(3):
        stamps.zero_()
        assert lib.me_debug_set_att_stamps(C.c_void_p(stamps.data_ptr())) == 0
        lib.me_op_attention(h, ptr(qkv), ptr(out), W, 577, 16)
        ctx.synchronize()
    lib.me_debug_set_att_stamps(None)
    us = timed()
    s = stamps.cpu().numpy().reshape(-1, 4, 8).astype(np.float64)
    act = s[:, :, 7] > 0
    a = s[act]
    print(f"== kernel v{ver} minw {minw or 4} ablation {mode}: {us:.1f} us (stamped build, events around a launch without a stamp buffer)"
          f"   workgroups {int((s[:, :, 6].sum(axis=1) > 0).sum())} active waves {int(act.sum())}", flush=True)
    for i, nm in enumerate(NAMES[ver]):
        print(f"  {nm:24s} mean {a[:, i].mean():9.0f} clocks per wave  ({a[:, i].mean() / 9.0:7.0f} per tile)  share {a[:, i].sum() / a[:, 6].sum():.3f}")
    if ver == "2":
        print(f"  prologue (to first tile)  mean {a[:, 7].mean():9.0f} clocks per wave")
    print("  unaccounted (prologue, tail key, epilogue) share %.3f" % (1 - a[:, :6].sum() / a[:, 6].sum()))
    inact = s[(~act) & (s[:, :, 6] > 0)]
    if len(inact):
        print(f"  waves without queries: {len(inact)}, mean lifetime {inact[:, 6].mean():.0f} clocks")
lib.me_debug_set_att_mode(0)
