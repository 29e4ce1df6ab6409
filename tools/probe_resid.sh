set -e
# proj / fc2 stand-alone at M = 21760: usage  [ME_GEMM_AG1=2] bash tools/probe_resid.sh "0 5"
for op in proj fc2; do for cfg in ${1:-0 5}; do
python - <<P
import sys, os, ctypes as C, math, torch
sys.path.insert(0, os.getcwd())
import matrix_eyes_amd as m
ctx = m.Context(0, "f16", m.ModelConfig.tiny()); lib, h = ctx.lib, ctx.handle
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
M = 21760; N, K = {"proj": (1024, 1024), "fc2": (1024, 4096)}["$op"]
a = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).half()
bias = torch.randn(N, device="cuda"); gamma = torch.rand(N, device="cuda") * 0.1; x = torch.randn(M, N, device="cuda")
p = lambda t: C.c_void_p(t.data_ptr())
f = lambda: lib.me_op_linear_residual(h, M, N, K, p(a), p(w), p(bias), p(gamma), p(x), $cfg)
for _ in range(5): assert f() == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(50): f()
e1.record(); e1.synchronize()
ms = e0.elapsed_time(e1) / 50
print("$op cfg $cfg AG1=%s: %.1f us  %.0f TFLOP/s" % (os.environ.get("ME_GEMM_AG1", "0"), ms * 1e3, 2.0 * M * N * K / ms / 1e9))
P
done; done
