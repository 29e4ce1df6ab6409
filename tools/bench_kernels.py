"""Kernel micro-benchmarks on one GPU (development tool): times the GEMM tile configurations at
the shapes of the 1536x1536 forward pass, the attention kernel and the 3x3 implicit-GEMM conv."""
import ctypes as C
import json
import math
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import matrix_eyes_amd as m


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters  # ms


def main():
    dtype = sys.argv[1] if len(sys.argv) > 1 else "f16"
    t16 = {"f16": torch.float16, "bf16": torch.bfloat16}[dtype]
    ctx = m.Context(0, dtype, m.ModelConfig.tiny())
    stream = torch.cuda.Stream()   # a real stream: handle 0 would select the context's own stream
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    lib, h = ctx.lib, ctx.handle
    res = []
    M = 35 * 577
    for (N, K, name) in [(3072, 1024, "qkv"), (1024, 1024, "proj"), (4096, 1024, "fc1"), (1024, 4096, "fc2")]:
        for Mx in (577, M, 21760, 4 * M):   # 21760 = the merged rows of the three ViTs at one image
            a = torch.randn(Mx, K, device="cuda").to(t16)
            w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(t16)
            bias = torch.randn(N, device="cuda")
            out16 = torch.empty(Mx, N, dtype=t16, device="cuda")
            x32 = torch.randn(Mx, N, device="cuda")
            gamma = torch.rand(N, device="cuda")
            for cfg in range(lib.me_op_gemm_config_count()):
                if name in ("proj", "fc2"):
                    f = lambda: lib.me_op_linear_residual(h, Mx, N, K, ptr(a), ptr(w), ptr(bias), ptr(gamma), ptr(x32), cfg)
                else:
                    act = 1 if name == "fc1" else 0
                    f = lambda: lib.me_op_linear(h, Mx, N, K, ptr(a), ptr(w), ptr(bias), ptr(out16), None, act, cfg)
                ms = timeit(f)
                tf = 2.0 * Mx * N * K / ms / 1e9
                res.append(dict(op=name, M=Mx, N=N, K=K, cfg=lib.me_op_gemm_config_name(cfg).decode(), ms=round(ms, 4), tflops=round(tf, 1)))
                print(res[-1], flush=True)
            del a, w, out16, x32
    # attention
    for W in (35, 140):
        qkv = (torch.randn(W * 577, 3072, device="cuda")).to(t16)
        out = torch.empty(W * 577, 1024, dtype=t16, device="cuda")
        ms = timeit(lambda: lib.me_op_attention(h, ptr(qkv), ptr(out), W, 577, 16))
        fl = W * 16 * 2 * 2 * 577 * 577 * 64
        res.append(dict(op="attention", windows=W, ms=round(ms, 4), tflops=round(fl / ms / 1e9, 1)))
        print(res[-1], flush=True)
    # layernorm
    x = torch.randn(M, 1024, device="cuda")
    wv = torch.ones(1024, device="cuda")
    y16 = torch.empty(M, 1024, dtype=t16, device="cuda")
    ms = timeit(lambda: lib.me_op_layernorm(h, ptr(x), ptr(wv), ptr(wv), ptr(y16), None, M, 1024, 1e-5))
    res.append(dict(op="layernorm", rows=M, ms=round(ms, 4), gbps=round(M * 1024 * 6 / ms / 1e6, 1)))
    print(res[-1], flush=True)
    # 3x3 conv 256->256 at 768^2 (decoder fusion[0]) and 384^2
    for Hh in (768, 384):
        xb = torch.randn(1, Hh + 2, Hh + 2, 256, device="cuda").to(t16)
        w = (torch.randn(256, 9 * 256, device="cuda") / 48).to(t16)
        bias = torch.randn(256, device="cuda")
        out16 = torch.zeros(1, Hh + 2, Hh + 2, 256, dtype=t16, device="cuda")
        r32 = torch.randn(Hh * Hh, 256, device="cuda")
        o32 = torch.empty(Hh * Hh, 256, device="cuda")
        for cfg in range(lib.me_op_gemm_config_count()):
            ms = timeit(lambda: lib.me_op_conv2d(h, ptr(xb), 1, Hh, Hh, 256, ptr(w), 256, 3, 1, ptr(bias), ptr(r32), None, ptr(o32), ptr(out16), 1, 2, 0, cfg), iters=5)
            fl = 2.0 * Hh * Hh * 2304 * 256
            res.append(dict(op="conv3x3", H=Hh, cfg=lib.me_op_gemm_config_name(cfg).decode(), ms=round(ms, 4), tflops=round(fl / ms / 1e9, 1)))
            print(res[-1], flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open(f"gpurun_out/bench_kernels_{dtype}.json", "w"), indent=1)


if __name__ == "__main__":
    main()
