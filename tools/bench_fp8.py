"""Kernel micro-benchmark (development tool): the MX fp8 GEMM against the 16-bit two-group kernel at the merged-row
shapes of the 1536x1536 step (M = 21760), all three epilogues.  python tools/bench_fp8.py"""
import ctypes as C, math, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import matrix_eyes_amd as m
from bench_kernels import ptr, timeit

ctx = m.Context(0, "f16", m.ModelConfig.tiny())
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); ctx.set_stream(stream.cuda_stream)
lib, h = ctx.lib, ctx.handle
res = []
M = 21760
for (N, K, name) in [(3072, 1024, "qkv"), (4096, 1024, "fc1"), (1024, 4096, "fc2")]:
    a = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).half()
    bias = torch.randn(N, device="cuda"); gamma = torch.rand(N, device="cuda"); x32 = torch.randn(M, N, device="cuda")
    out16 = torch.empty(M, N, dtype=torch.float16, device="cuda")
    a8 = torch.empty(M, K, dtype=torch.uint8, device="cuda"); asc = torch.zeros(M * K // 32, dtype=torch.uint8, device="cuda")
    w8 = torch.empty(N, K, dtype=torch.uint8, device="cuda"); wsc = torch.zeros(N * K // 32, dtype=torch.uint8, device="cuda")
    o8 = torch.empty(M, N, dtype=torch.uint8, device="cuda"); osc = torch.zeros(M * N // 32, dtype=torch.uint8, device="cuda")
    ctx._check(lib.me_op_quantize_fp8(h, ptr(a), M, K, 0, ptr(a8), ptr(asc)))
    ctx._check(lib.me_op_quantize_fp8(h, ptr(w), N, K, 1, ptr(w8), ptr(wsc)))
    if name == "qkv":
        f8 = lambda: lib.me_op_linear_fp8(h, M, N, K, ptr(a8), ptr(asc), ptr(w8), ptr(wsc), ptr(bias), ptr(out16), None, None, None, None)
        f16 = lambda: lib.me_op_linear(h, M, N, K, ptr(a), ptr(w), ptr(bias), ptr(out16), None, 0, 0)
    elif name == "fc1":
        f8 = lambda: lib.me_op_linear_fp8(h, M, N, K, ptr(a8), ptr(asc), ptr(w8), ptr(wsc), ptr(bias), None, ptr(o8), ptr(osc), None, None)
        f16 = lambda: lib.me_op_linear(h, M, N, K, ptr(a), ptr(w), ptr(bias), ptr(out16), None, 1, 0)
    else:
        f8 = lambda: lib.me_op_linear_fp8(h, M, N, K, ptr(a8), ptr(asc), ptr(w8), ptr(wsc), ptr(bias), None, None, None, ptr(gamma), ptr(x32))
        f16 = lambda: lib.me_op_linear_residual(h, M, N, K, ptr(a), ptr(w), ptr(bias), ptr(gamma), ptr(x32), 0)
    for tag, f in (("fp8", f8), ("f16", f16)):
        ms = timeit(f, iters=20)
        res.append(dict(op=name, M=M, N=N, K=K, kernel=tag, ms=round(ms, 4), tflops=round(2.0 * M * N * K / ms / 1e9, 1)))
        print(res[-1], flush=True)
x = torch.randn(M, 1024, device="cuda"); wv = torch.ones(1024, device="cuda")
y8 = torch.empty(M, 1024, dtype=torch.uint8, device="cuda"); ys = torch.zeros(M * 32, dtype=torch.uint8, device="cuda")
y16 = torch.empty(M, 1024, dtype=torch.float16, device="cuda")
ms = timeit(lambda: lib.me_op_layernorm_fp8(h, ptr(x), ptr(wv), ptr(wv), ptr(y8), ptr(ys), M, 1024, 1e-5)); print("layernorm_fp8 ms", ms)
ms = timeit(lambda: lib.me_op_layernorm(h, ptr(x), ptr(wv), ptr(wv), ptr(y16), None, M, 1024, 1e-5)); print("layernorm f16 ms", ms)
json.dump(res, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r02_fp8_microbench.json"), "w"), indent=1)
