"""Development: per-wave phase clocks of head_halo_kernel (needs a -DME_HEAD_STAMPS build of the library:
make -C matrix-eyes_amd/csrc BUILD=../build_stamps OUT=../../build_ab/libheadstamps.so CXXFLAGS="... -DME_HEAD_STAMPS";
run with MATRIX_EYES_HIP_LIB pointing at it)."""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import matrix_eyes_amd as m
from tools.bench_kernels import ptr

ctx = m.Context(0, "f16", m.ModelConfig.tiny())
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
lib, h = ctx.lib, ctx.handle
lib.me_debug_set_head_stamps.argtypes = [C.c_void_p]
B, S, Cin, Cmid = 1, 1536, 128, 32
xb = torch.zeros(B, S + 2, S + 2, Cin, dtype=torch.float16, device="cuda")
xb[:, 1:S + 1, 1:S + 1] = torch.randn(B, S, S, Cin, device="cuda").half()
w = (torch.randn(Cmid, 9 * Cin, device="cuda") / math.sqrt(9 * Cin)).half()
bias, w2, b2 = torch.randn(Cmid, device="cuda") * 0.3, torch.randn(Cmid, device="cuda") / 6, torch.tensor([0.4], device="cuda")
fn = torch.ones(B, device="cuda")
out = torch.empty(B * S * S, device="cuda")
stamps = torch.zeros(256 * 8 * 8, dtype=torch.int64, device="cuda")
call = lambda: lib.me_op_head_final(h, ptr(xb), B, S, S, Cin, ptr(w), Cmid, ptr(bias), ptr(w2), ptr(b2), ptr(fn), 1e-4, 1e4, ptr(out), -1)
for _ in range(3):
    assert call() == 0
torch.cuda.synchronize()
assert lib.me_debug_set_head_stamps(C.c_void_p(stamps.data_ptr())) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); call(); e1.record(); torch.cuda.synchronize()
lib.me_debug_set_head_stamps(None)
s = stamps.cpu().numpy().reshape(256, 8, 8).astype(np.float64)
print(f"launch {e0.elapsed_time(e1) * 1e3:.1f} us; tiles per workgroup {s[:, 0, 5].mean():.1f}")
names = ["halo wait + barrier", "MFMAs (+ next halo's requests)", "partial sums + barrier", "finishing (kh = 0) / idle"]
for khv, nm in ((0, "kh = 0 waves"), (1, "kh = 1 waves")):
    a = s[:, 4 * khv:4 * khv + 4, :].reshape(-1, 8)
    print(nm, f"whole kernel {a[:, 4].mean():.0f} clocks (100 MHz x ?: s_memtime)")
    for i, n in enumerate(names):
        print(f"   {n:34s} {a[:, i].mean() / a[:, 5].mean():8.0f} clocks per tile   share {a[:, i].sum() / a[:, 4].sum():.3f}")
