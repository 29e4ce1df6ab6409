"""development: time one GEMM shape per config under ME_GEMM_DEBUG (0 full, 1 loads only, 2 compute only)"""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import matrix_eyes_amd as m
from tools.bench_kernels import timeit, ptr
ctx = m.Context(0, "f16", m.ModelConfig.tiny())
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); ctx.set_stream(stream.cuda_stream)
lib, h = ctx.lib, ctx.handle
M = 35 * 577
for (N, K, name) in [(3072, 1024, "qkv"), (1024, 4096, "fc2")]:
    a = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).half()
    bias = torch.randn(N, device="cuda"); out16 = torch.empty(M, N, dtype=torch.float16, device="cuda")
    for cfg in (0, 1, 2, 3):
        ms = timeit(lambda: lib.me_op_linear(h, M, N, K, ptr(a), ptr(w), ptr(bias), ptr(out16), None, 0, cfg))
        print(os.environ.get("ME_GEMM_DEBUG", "0"), name, lib.me_op_gemm_config_name(cfg).decode(), round(ms, 4), "ms", round(2.0 * M * N * K / ms / 1e9, 1), "TF-equiv", flush=True)
