"""Runs ONE kernel shape repeatedly so that rocprofv3 --pmc / --kernel-trace can be pointed at it.
usage: gemm_probe.py <op> <cfg> [iters]   op in qkv|proj|fc1|fc2|conv768|attn"""
import ctypes as C
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import matrix_eyes_amd as m


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def main():
    op, cfg = sys.argv[1], int(sys.argv[2])
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    ctx = m.Context(0, "f16", m.ModelConfig.tiny())
    lib, h = ctx.lib, ctx.handle
    t16 = torch.float16
    M = int(os.environ.get("PROBE_M", 21760))   # the merged rows of the three ViTs at one image (35 * 577 = one ViT)
    shapes = {"qkv": (3072, 1024), "proj": (1024, 1024), "fc1": (4096, 1024), "fc2": (1024, 4096)}
    if op in shapes:
        N, K = shapes[op]
        a = torch.randn(M, K, device="cuda").to(t16)
        w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(t16)
        bias = torch.randn(N, device="cuda")
        out16 = torch.empty(M, N, dtype=t16, device="cuda")
        x32 = torch.randn(M, N, device="cuda")
        gamma = torch.rand(N, device="cuda")
        if op in ("proj", "fc2") and cfg == 10:
            # the step's form: the 352-row tile with the next sublayer's LayerNorm in the residual epilogue
            lw = torch.ones(N, device="cuda"); lb = torch.zeros(N, device="cuda")
            xn = torch.empty(M, N, dtype=t16, device="cuda")
            arr = lambda t: (C.c_void_p * 3)(t.data_ptr(), 0, 0)
            f = lambda: lib.me_op_linear_residual_layernorm(h, M, N, K, ptr(a), 0, 0, arr(w), arr(bias), arr(gamma), arr(lw),
                                                            arr(lb), 1e-5, ptr(x32), ptr(xn))
        elif op in ("proj", "fc2"):
            f = lambda: lib.me_op_linear_residual(h, M, N, K, ptr(a), ptr(w), ptr(bias), ptr(gamma), ptr(x32), cfg)
        else:
            f = lambda: lib.me_op_linear(h, M, N, K, ptr(a), ptr(w), ptr(bias), ptr(out16), None, 1 if op == "fc1" else 0, cfg)
    elif op in ("qkv8", "fc1_8", "fc2_8"):     # MX fp8 operands (csrc/gemm_fp8.hip)
        N, K = {"qkv8": (3072, 1024), "fc1_8": (4096, 1024), "fc2_8": (1024, 4096)}[op]
        a = torch.randn(M, K, device="cuda").to(t16)
        w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(t16)
        bias = torch.randn(N, device="cuda"); gamma = torch.rand(N, device="cuda"); x32 = torch.randn(M, N, device="cuda")
        out16 = torch.empty(M, N, dtype=t16, device="cuda")
        a8 = torch.empty(M, K, dtype=torch.uint8, device="cuda"); asc = torch.zeros(M * K // 32, dtype=torch.uint8, device="cuda")
        w8 = torch.empty(N, K, dtype=torch.uint8, device="cuda"); wsc = torch.zeros(N * K // 32, dtype=torch.uint8, device="cuda")
        o8 = torch.empty(M, N, dtype=torch.uint8, device="cuda"); osc = torch.zeros(M * N // 32, dtype=torch.uint8, device="cuda")
        assert lib.me_op_quantize_fp8(h, ptr(a), M, K, 0, ptr(a8), ptr(asc)) == 0
        assert lib.me_op_quantize_fp8(h, ptr(w), N, K, 1, ptr(w8), ptr(wsc)) == 0
        if op == "qkv8":
            f = lambda: lib.me_op_linear_fp8(h, M, N, K, ptr(a8), ptr(asc), ptr(w8), ptr(wsc), ptr(bias), ptr(out16), None, None, None, None)
        elif op == "fc1_8":
            f = lambda: lib.me_op_linear_fp8(h, M, N, K, ptr(a8), ptr(asc), ptr(w8), ptr(wsc), ptr(bias), None, ptr(o8), ptr(osc), None, None)
        else:
            f = lambda: lib.me_op_linear_fp8(h, M, N, K, ptr(a8), ptr(asc), ptr(w8), ptr(wsc), ptr(bias), None, None, None, ptr(gamma), ptr(x32))
    elif op == "conv768":
        Hh = 768
        xb = torch.randn(1, Hh + 2, Hh + 2, 256, device="cuda").to(t16)
        w = (torch.randn(256, 9 * 256, device="cuda") / 48).to(t16)
        bias = torch.randn(256, device="cuda")
        out16 = torch.zeros(1, Hh + 2, Hh + 2, 256, dtype=t16, device="cuda")
        r32 = torch.randn(Hh * Hh, 256, device="cuda")
        o32 = torch.empty(Hh * Hh, 256, device="cuda")
        f = lambda: lib.me_op_conv2d(h, ptr(xb), 1, Hh, Hh, 256, ptr(w), 256, 3, 1, ptr(bias), ptr(r32), None, ptr(o32), ptr(out16), 1, 2, 0, cfg)
    elif op == "attn":
        M = 35 * 577
        qkv = torch.randn(M, 3072, device="cuda").to(t16)
        out = torch.empty(M, 1024, dtype=t16, device="cuda")
        f = lambda: lib.me_op_attention_prescaled(h, ptr(qkv), ptr(out), 35, 577, 16)   # the step's kernel (Q pre-scaled)
    else:
        raise SystemExit("unknown op")
    for _ in range(iters):
        rc = f()
        assert rc == 0, rc
    ctx.synchronize()


if __name__ == "__main__":
    main()
