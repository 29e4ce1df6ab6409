"""fc1 (GELU epilogue) on the 352-row tile at the step's shape: the batched GELU against the four-values-at-a-time form
(ME_GELU_BATCH=0), one process, interleaved rounds"""
import ctypes as C, math, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import matrix_eyes_amd as m
ctx = m.Context(0, "f16", m.ModelConfig.tiny())
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
M, N, K = 21760, 4096, 1024
a = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).half()
bias = torch.randn(N, device="cuda"); out = torch.empty(M, N, dtype=torch.float16, device="cuda")
p = lambda t: C.c_void_p(t.data_ptr())
def run(env):
    os.environ["ME_GELU_BATCH"] = env
    assert ctx.lib.me_op_linear(ctx.handle, M, N, K, p(a), p(w), p(bias), p(out), None, 1, 10) == 0
outs = {}
for env in ("0", "1"):
    run(env); torch.cuda.synchronize(); outs[env] = out.clone()
print("bit-identical:", bool(torch.equal(outs["0"], outs["1"])))
ts = {"0": [], "1": []}
for r in range(12):
    for env in ("0", "1"):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): run(env)
        e1.record(); torch.cuda.synchronize(); ts[env].append(e0.elapsed_time(e1) * 1e3 / 4)
for env in ("0", "1"):
    med = statistics.median(ts[env])
    print(f"ME_GELU_BATCH={env}: median {med:.1f} us  min {min(ts[env]):.1f} us  {2.0 * M * N * K / med / 1e6:.0f} TFLOP/s")
